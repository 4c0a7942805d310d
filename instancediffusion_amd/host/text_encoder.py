"""Host mirror of ``ldm/modules/encoders/modules.py:FrozenCLIPEmbedder``.

The text encoder runs ONCE per prompt, before the sampling path; it is a neighbour of the path, not part of it, so it
stays on Hugging Face ``transformers`` (PyTorch-ROCm eager) exactly as in the reference.  Difference forced by the
offline build environment: nothing is downloaded.  The CLIP-L/14 text transformer is built from its (fixed) config so
that the checkpoint's ``['text_encoder']`` sub-dict -- which holds all its weights -- loads into it
(``utils/checkpoint.py:246``); the BPE tokenizer needs the ``vocab.json`` / ``merges.txt`` of
``openai/clip-vit-large-patch14`` in the local HF cache or under ``$IDF_CLIP_PATH`` and raises a clear error otherwise.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

CLIP_L14_TEXT = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, projection_dim=768,
                     num_hidden_layers=12, num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                     layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=49406, eos_token_id=49407)


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenCLIPEmbedder(AbstractEncoder):
    """encoders/modules.py:144-172: CLIP text transformer -> last_hidden_state [B, 77, 768] (+ pooler_output)."""

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77):
        super().__init__()
        from transformers import CLIPTextConfig, CLIPTextModel
        self.version = os.environ.get("IDF_CLIP_PATH", version)
        self.transformer = CLIPTextModel(CLIPTextConfig(**CLIP_L14_TEXT))    # weights come from the checkpoint
        self.device = device
        self.max_length = max_length
        self._tokenizer = None
        self.freeze()

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from transformers import CLIPTokenizer
            msg = (f"CLIP BPE vocabulary for '{self.version}' not found locally (no network here): put vocab.json and "
                   f"merges.txt of openai/clip-vit-large-patch14 in a directory and set IDF_CLIP_PATH to it")
            try:
                tok = CLIPTokenizer.from_pretrained(self.version, local_files_only=True)
            except Exception as e:
                raise RuntimeError(msg) from e
            if len(tok) < CLIP_L14_TEXT["vocab_size"]:      # transformers >= 5 builds an EMPTY tokenizer without files
                raise RuntimeError(msg)
            self._tokenizer = tok
        return self._tokenizer

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accept both key layouts of ``CLIPTextModel``: ``transformer.text_model.*`` (transformers 4.x, the layout the
        reference checkpoints were saved with) and ``transformer.*`` (transformers >= 5).  The reference loads this
        sub-dict with strict=False (utils/checkpoint.py:246), which would silently load NOTHING on a layout mismatch --
        so a call that matches no weight at all raises."""
        mine = set(self.state_dict().keys())
        nested_here = any(k.startswith("transformer.text_model.") for k in mine)
        remapped = {}
        for k, v in state_dict.items():
            if k.startswith("transformer.text_model.") and not nested_here:
                k = "transformer." + k[len("transformer.text_model."):]
            elif k.startswith("transformer.") and nested_here and not k.startswith("transformer.text_model."):
                k = "transformer.text_model." + k[len("transformer."):]
            remapped[k] = v
        if len(state_dict) and not (set(remapped) & mine):
            raise RuntimeError("text_encoder state dict matches none of the CLIP text transformer's parameters")
        return super().load_state_dict(remapped, strict=strict, **kw)

    def freeze(self):
        self.transformer = self.transformer.eval()
        for param in self.parameters():
            param.requires_grad = False

    def to(self, *args, **kwargs):
        r = super().to(*args, **kwargs)
        if args and isinstance(args[0], (str, torch.device)):
            self.device = args[0]
        return r

    @torch.no_grad()
    def forward(self, text, return_pooler_output=False):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        outputs = self.transformer(input_ids=enc["input_ids"].to(self.device))
        z = outputs.last_hidden_state
        return (z, outputs.pooler_output) if return_pooler_output else z

    def encode(self, text, return_pooler_output=False):
        return self(text, return_pooler_output)
