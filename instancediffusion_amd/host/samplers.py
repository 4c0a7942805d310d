"""PLMS sampler and the Multi-instance Sampler (MIS) for the MI355X engine.

Host mirror of ``ldm/models/diffusion/plms.py`` (PLMSSampler) and ``plms_instance.py`` (PLMSSamplerInst): same
constructor / ``sample()`` API, same schedule, same update rule and quirks, different execution plan:

  * the conditional and unconditional (classifier-free guidance) evaluations of a step run as ONE batched UNet
    forward; in MIS phase 1 all N+1 instance trajectories (x images) advance together in one batch -- the reference
    runs those 2(N+1) forwards per step serially (plms_instance.py:86-104);
  * everything step-invariant lives in ``engine.Cond`` objects built once per ``sample()`` call;
  * latents, eps history and the PLMS/CFG arithmetic stay on the GPU in fp32 (fused kernels);
  * with ``torch.distributed`` initialised (one process per GPU, RCCL), MIS phase 1 is sharded over
    (instance, image) work units, the instance latents are recombined by ONE all-gather of the unit latents each rank owns
    (scattered by index into the fixed [instance][image] stack: pure data movement) and merged by the
    same ``idf_mis_merge`` call as on one rank -- outputs are bit-identical at every world size -- and phase 2 is
    sharded over images.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .diffusion import make_ddim_timesteps


def guided_uc_shared(uc: torch.Tensor) -> bool:
    """Are all rows of the unconditional context the same (stride-0 broadcast, or equal values)?  The stride test costs
    nothing; the value test is one full-tensor compare and a host sync, run ONCE per ``sample()`` call on the live tensor.
    (Round 4 memoised it under (address, shape, strides, version) -- a key that does not identify the tensor's CONTENT: a fresh
    ``uc`` of another call can reuse the address at version 0 and hit a stale True, which silently broadcast row 0's negative
    prompt to every image; ``_version`` also raises under inference_mode.  ADVICE r4.)"""
    if uc.shape[0] <= 1:
        return False
    if uc.stride(0) == 0:
        return True
    return bool((uc[1:] == uc[:1]).all())


class _PLMSBase(object):
    def __init__(self, diffusion, model, schedule="linear", alpha_generator_func=None, set_alpha_scale=None):
        super().__init__()
        self.diffusion = diffusion
        self.model = model
        self.device = diffusion.betas.device
        self.ddpm_num_timesteps = diffusion.num_timesteps
        self.schedule = schedule
        self.alpha_generator_func = alpha_generator_func
        self.set_alpha_scale = set_alpha_scale

    # ---- schedule (plms.py:25-62; util.py:55-83) ------------------------------------------------------
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=False):
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        ac = self.diffusion.alphas_cumprod.detach().to(torch.float32).cpu()
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        # float32 values, exactly as the reference's float32 buffers (make_ddim_sampling_parameters, eta = 0)
        self.ddim_alphas = ac[self.ddim_timesteps].numpy()
        self.ddim_alphas_prev = np.asarray([ac[0].item()] + ac[self.ddim_timesteps[:-1]].tolist(), dtype=np.float32)
        self.ddim_sigmas = np.zeros_like(self.ddim_alphas)
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1.0 - ac[self.ddim_timesteps]).numpy()

    # ---- engine plumbing --------------------------------------------------------------------------------
    @property
    def engine(self):
        return self.model.engine

    def _cond(self, inp: Dict) -> "object":
        g = inp["grounding_input"] if "grounding_input" in inp else self.model.grounding_tokenizer_input.get_null_input()
        return self.engine.prepare_cond(inp["context"], g)

    def _uncond(self, uc: torch.Tensor):
        return self.engine.prepare_cond(uc, self.model.grounding_tokenizer_input.get_null_input(batch=uc.shape[0]))

    def _check_first_conv(self, alphas):
        """The alpha schedule reaches 0 (default alpha_type [0.8, 0, 0.2]: at step 40 of 50) -> the model will swap its first
        conv for Stable Diffusion's (openaimodel.py:469-480) and needs that 48-KB file: find out NOW, not 40 steps in."""
        if alphas is not None and any(a == 0 for a in alphas) and hasattr(self.model, "check_first_conv_available"):
            self.model.check_first_conv_available()

    def _apply_alpha(self, alphas, i):
        """plms.py:90-94: per-step gate + first-conv swap."""
        if alphas is not None:
            self.set_alpha_scale(self.model, alphas[i])
            if alphas[i] == 0:
                self.model.restore_first_conv_from_SD()
            self.engine.sync_fuser_scale_from_modules()

    def _eps(self, x, step: int, cond_pair, n: int, guided: bool, guidance_scale: float):
        """Guided eps for n trajectories: one batched forward of [cond | uncond] (plms.py:121-127)."""
        eng = self.engine
        if guided:
            # the n distinct rows only: the engine writes both halves of its static [cond | uncond] input itself (the
            # paired invariant is structural, not a promise checked once per launch configuration)
            t = torch.full((n,), float(step), device=x.device, dtype=torch.float32)
            e2 = eng.forward_cond(x, t, cond_pair, out=eng.buf("smp.eps2", (2 * n,) + tuple(x.shape[1:]), torch.float32), paired=True)
            return eng.ops.cfg_combine(e2[:n], e2[n:], guidance_scale, eng.ops.empty(x.shape, torch.float32))
        t = torch.full((n,), float(step), device=x.device, dtype=torch.float32)
        return eng.forward_cond(x, t, cond_pair)

    def _plms_step(self, x, old_eps: list, index: int, step: int, step_next: int, cond_pair, n, guided, gs,
                   e_t_first=None):
        """p_sample_plms (plms.py:117-167).  Returns (x_prev, e_t).  ``e_t_first``: guided eps of the first
        evaluation when the caller already computed it (MIS start-of-trajectory de-duplication)."""
        ops = self.engine.ops
        a_t, a_prev = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index])
        s1m = float(self.ddim_sqrt_one_minus_alphas[index])
        e_t = e_t_first if e_t_first is not None else self._eps(x, step, cond_pair, n, guided, gs)
        new = ops.empty(x.shape, torch.float32)
        if len(old_eps) == 0:
            x_pred = ops.plms_update(x, e_t, [], None, 0, a_t, a_prev, s1m, ops.empty(x.shape, torch.float32))
            e_next = self._eps(x_pred, step_next, cond_pair, n, guided, gs)
            ops.plms_update(x, e_t, [], e_next, 1, a_t, a_prev, s1m, new)
        else:
            ops.plms_update(x, e_t, old_eps, None, min(len(old_eps), 3) + 1, a_t, a_prev, s1m, new)
        return new, e_t

    @staticmethod
    def _push(old_eps: list, e_t):
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)


class PLMSSampler(_PLMSBase):
    """plms.py:9-167."""

    @torch.no_grad()
    def sample(self, S, shape, input, uc=None, guidance_scale=1, mask=None, x0=None):
        self.make_schedule(ddim_num_steps=S)
        return self.plms_sampling(shape, input, uc, guidance_scale, mask=mask, x0=x0)

    @torch.no_grad()
    def plms_sampling(self, shape, input, uc=None, guidance_scale=1, mask=None, x0=None):
        eng = self.engine
        dev = eng.device
        b = shape[0]
        img = input["x"]
        if img is None:
            img = torch.randn(shape, device=dev)
            input["x"] = img
        img = img.to(dev, torch.float32)
        guided = uc is not None and guidance_scale != 1
        cond = self._cond(input)
        if guided:
            bank = type(cond).cat([cond, self._uncond(uc)])
            pair = eng.gather_cond(bank, torch.arange(2 * b, device=dev))
        else:
            pair = cond
        time_range = np.flip(self.ddim_timesteps)
        total = self.ddim_timesteps.shape[0]
        alphas = self.alpha_generator_func(len(time_range)) if self.alpha_generator_func is not None else None
        self._check_first_conv(alphas)
        old_eps: list = []
        for i, step in enumerate(time_range):
            self._apply_alpha(alphas, i)
            index = total - i - 1
            step_next = int(time_range[min(i + 1, len(time_range) - 1)])
            if mask is not None:
                assert x0 is not None
                ts = torch.full((b,), int(step), device=dev, dtype=torch.long)
                img_orig = self.diffusion.q_sample(x0.to(dev), ts)
                img = img_orig * mask + (1. - mask) * img
            img, e_t = self._plms_step(img, old_eps, index, int(step), step_next, pair, b, guided, guidance_scale)
            input["x"] = img
            input["timesteps"] = torch.full((b,), int(step), device=dev, dtype=torch.long)
            self._push(old_eps, e_t)
        return img


class PLMSSamplerInst(_PLMSBase):
    """Multi-instance Sampler, plms_instance.py:7-212."""

    def __init__(self, diffusion, model, schedule="linear", alpha_generator_func=None, set_alpha_scale=None, mis=0.0,
                 crop_and_paste_latents=False, shard_across_ranks: Optional[bool] = None, max_units: int = 128,
                 unit_sharding: str = "auto"):
        super().__init__(diffusion, model, schedule, alpha_generator_func, set_alpha_scale)
        self.mis = mis
        self.crop_and_paste_latents = crop_and_paste_latents      # hard-coded False in the reference (:128)
        self.shard_across_ranks = shard_across_ranks
        assert unit_sharding in ("auto", "image", "instance")
        self.unit_sharding = unit_sharding
        # phase-1 (instance, image) units per batched forward: 128 units x [cond | uncond] = 256-row forwards (1.230 ms per row
        # against 1.250 at 128 rows and 1.27 at 64, graph-replayed: profiles/r06_replay_256.log; bench + 1.2 %, r06_maxunits_ab.log)
        self.max_units = max_units

    @torch.no_grad()
    def sample(self, S, shape, input, uc=None, guidance_scale=1, mask=None, x0=None):
        self.make_schedule(ddim_num_steps=S)
        return self.plms_sampling(shape, input, uc, guidance_scale, mask=mask, x0=x0)

    @staticmethod
    def units_per_forward(max_units: int, h: int, w: int) -> int:
        """(instance, image) units per phase-1 forward: ``max_units`` at latents up to 64 x 64; beyond (C4: 96 x 96) the same TOKEN
        budget per forward (2 max_units x 4096 rows of the 64^2 level: 2^20 at the default, the width every kernel's index ranges
        are tested at), but never fewer than the 64 units the default of rounds 3-6 formed at every size."""
        return min(max_units, max(64, max_units * 4096 // (h * w)))

    @staticmethod
    def chunk_sizes(n_units: int, per_fwd: int):
        """Phase-1 forwards of a rank's n_units: full ones of per_fwd units, then the remainder -- cut at a multiple of 64 units when
        it is larger than that: 128-row multiples fill the persistent kernels' tile grids (2 / 4 full rounds of the 32^2 level's
        256-row tiles on 256 CUs), widths in between do not (72 units as ONE 144-row forward ran the 8-image bench leg 3 % slower
        than 64 + 8: profiles/r06_bench_final.json against r06_bench_final_maxunits64_*.json)."""
        out, left = [], n_units
        while left > 0:
            n = per_fwd if left >= per_fwd else (left if left <= 64 or left % 64 == 0 else left // 64 * 64)
            out.append(n)
            left -= n
        return out

    # ---- distributed helpers ----------------------------------------------------------------------------
    def _dist(self):
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if self.shard_across_ranks is False:
            on = False
        return (dist, dist.get_rank(), dist.get_world_size()) if on else (None, 0, 1)

    @torch.no_grad()
    def plms_sampling(self, shape, input_all, uc=None, guidance_scale=1, mask=None, x0=None):
        eng = self.engine
        ops = eng.ops
        dev = eng.device
        dist, rank, world = self._dist()
        B = shape[0]
        latent_size = shape[2]
        img = input_all[0]["x"]
        if img is None:
            img = torch.randn(shape, device=dev)
            if dist is not None:
                dist.broadcast(img, 0)
            for inp in input_all:
                inp["x"] = img
        n_all = len(input_all)
        guided = uc is not None and guidance_scale != 1
        time_range = np.flip(self.ddim_timesteps)
        total = self.ddim_timesteps.shape[0]
        alphas = self.alpha_generator_func(len(time_range)) if self.alpha_generator_func is not None else None
        self._check_first_conv(alphas)
        mis_step = int(total * self.mis)

        # ---------------- work split (decided BEFORE any conditioning is built) ---------------------------------
        # Phase 1 = N+1 independent trajectories per image (plms_instance.py:86-104); work unit = (instance j, image b).  "image" sharding: owner = b % world -- every trajectory of an image lives on
        # the rank that also runs its phase 2, the first-evaluation hoist touches only this rank's images and per-rank work
        # is independent of the world size (weak scaling); used when the images divide evenly over the ranks.
        # "instance" sharding: owner = (b + j) % world -- spreads the N+1 trajectories of FEW images (B < world, e.g. one
        # image on 8 GPUs) over the ranks; the owner of image b (rank b % world) still runs (0, b), whose eps history
        # continues into phase 2.  Either way the merge is at most one small all-gather.
        mode = self.unit_sharding
        if mode == "auto":
            mode = "image" if (B % world == 0) else "instance"
        if mode == "image":
            units = [(j, b) for j in range(n_all) for b in range(B) if b % world == rank]
        else:
            units = [(j, b) for j in range(n_all) for b in range(B) if (b + j) % world == rank]
        mine = [b for b in range(B) if b % world == rank]                     # images this rank owns in phase 2

        # Conditioning (UniFusion tokens + per-layer K/V caches) only for the (input, image) pairs THIS rank touches: input j
        # is prepared on the images of this rank's units (j, b) -- with either ownership rule that is B / world images per
        # input, so per-rank setup work and HBM stay constant as ranks are added (weak scaling); the unconditional branch
        # is prepared on the union.  One bank of all conditionings; batches are assembled by a single row gather straight
        # into the engine's static slot.
        need_j = [sorted({b for (jj, b) in units if jj == j} | (set(mine) if j == 0 else set())) for j in range(n_all)]
        need_u = sorted({b for (_, b) in units} | set(mine))
        # identical unconditional rows (one negative prompt broadcast over the batch, inference.py:303): build that
        # conditioning ONCE and let every image gather the same bank row -- with `instance` ownership need_u is all B
        # images of the global batch, and per-rank setup work must not grow with the world size
        uc_shared = bool(guided_uc_shared(uc)) if (uc is not None and len(need_u) > 1) else False
        row_of: Dict[tuple, int] = {}
        off = 0
        for j in range(n_all):
            for k, b in enumerate(need_j[j]):
                row_of[(j, b)] = off + k
            off += len(need_j[j])
        row_unc = {b: off + (0 if uc_shared else k) for k, b in enumerate(need_u)}

        def take(v, imgs):
            if not torch.is_tensor(v) or v.dim() == 0 or v.shape[0] != B or len(imgs) == B:
                return v
            if v.stride(0) == 0:                                               # batch-broadcast view stays a view
                return v[:1].expand(len(imgs), *v.shape[1:])
            return v.index_select(0, torch.tensor(imgs, device=v.device, dtype=torch.long))

        def local(inp, imgs):
            d = dict(inp)
            d["context"] = take(inp["context"], imgs)
            if "grounding_input" in inp:
                gin = inp["grounding_input"]
                d["grounding_input"] = {k: take(v, imgs) for k, v in gin.items()}
                if torch.is_tensor(gin.get("att_masks")) and len(imgs) != B:
                    # attention.py:200 decides mask / no mask on the WHOLE per-call tensor: take that decision here, on
                    # the un-sharded batch, so that a rank's row subset cannot flip it (results independent of world)
                    d["grounding_input"]["att_masks_any"] = bool((gin["att_masks"] > 0).any())
            else:
                d["grounding_input"] = self.model.grounding_tokenizer_input.get_null_input(batch=len(imgs))
            return d

        conds = [self._cond(local(input_all[j], need_j[j])) for j in range(n_all) if need_j[j]]
        cond_u = self._uncond(take(uc, need_u[:1] if uc_shared else need_u)) if (guided and need_u) else None
        parts = conds + ([cond_u] if cond_u is not None else [])
        bank = type(parts[0]).cat(parts) if parts else None

        def rows(units):
            r = [row_of[u] for u in units]
            if guided:
                r += [row_unc[b] for (_, b) in units]
            return torch.tensor(r, device=dev, dtype=torch.long)

        # Reference quirk: restore_first_conv_from_SD is never undone, so if alpha hits 0 inside phase 1 the LATER
        # instances would run their EARLY steps with the swapped conv.  Only then is the serial order observable;
        # reproduce it by advancing one instance at a time.
        serial = alphas is not None and any(alphas[i] == 0 for i in range(mis_step))
        per_fwd = self.units_per_forward(self.max_units, int(shape[2]), int(shape[3]))
        if serial:
            chunks = [[u for u in units if u[0] == j] for j in range(n_all)]
        else:
            chunks, k = [], 0
            for n in self.chunk_sizes(len(units), per_fwd):
                chunks.append(units[k:k + n])
                k += n
        x_units: Dict[tuple, torch.Tensor] = {}
        eps_units: Dict[tuple, list] = {}
        same_start = all(inp["x"] is input_all[0]["x"] or torch.equal(inp["x"], input_all[0]["x"]) for inp in input_all[1:])
        # Exact hoist of the very first evaluation: every instance trajectory of image b still holds the SAME latent (the
        # shared starting noise, inference.py:300-301) and the unconditional branch sees only (x, t, uc, null grounding),
        # so it is identical for the N+1 instances of an image.  Evaluate all conditional rows of this rank's units plus
        # ONE unconditional row per image up front, in full-width forwards, instead of one unconditional row per unit.
        first_idx: Dict[tuple, int] = {}
        first_unc: Dict[int, int] = {}
        E_first = None
        if guided and same_start and units and mis_step > 0 and not serial:
            self._apply_alpha(alphas, 0)
            imgs = sorted({b for (_, b) in units})
            row_ids = [row_of[u] for u in units] + [row_unc[b] for b in imgs]
            row_img = [b for (_, b) in units] + imgs
            first_idx = {u: k for k, u in enumerate(units)}
            first_unc = {b: len(units) + k for k, b in enumerate(imgs)}
            x0_all = input_all[0]["x"].to(dev, torch.float32)
            width = 2 * per_fwd
            parts = []
            for k in range(0, len(row_ids), width):
                r = torch.tensor(row_ids[k:k + width], device=dev, dtype=torch.long)
                xx = x0_all[torch.tensor(row_img[k:k + width], device=dev, dtype=torch.long)]
                tt = torch.full((xx.shape[0],), float(time_range[0]), device=dev, dtype=torch.float32)
                slot = eng.gather_cond(bank, r)
                parts.append(eng.forward_cond(xx, tt, slot, out=eng.buf("smp.eps_first", xx.shape, torch.float32)).clone())
            E_first = torch.cat(parts, 0)
        for chunk in chunks:
            if not chunk:
                continue
            n = len(chunk)
            pair = None
            x = torch.stack([input_all[j]["x"][b] for (j, b) in chunk]).to(dev, torch.float32)
            old: list = []
            for i, step in enumerate(time_range[:mis_step]):
                self._apply_alpha(alphas, i)
                index = total - i - 1
                step_next = int(time_range[min(i + 1, len(time_range) - 1)])
                e_first = None
                if i == 0 and E_first is not None:
                    ic = torch.tensor([first_idx[u] for u in chunk], device=dev, dtype=torch.long)
                    iu = torch.tensor([first_unc[b] for (_, b) in chunk], device=dev, dtype=torch.long)
                    e_first = ops.cfg_combine(E_first[ic].contiguous(), E_first[iu].contiguous(), guidance_scale,
                                              ops.empty(x.shape, torch.float32))
                if i == 0 or pair is None:
                    pair = eng.gather_cond(bank, rows(chunk))
                x, e_t = self._plms_step(x, old, index, int(step), step_next, pair, n, guided, guidance_scale,
                                         e_t_first=e_first)
                self._push(old, e_t)
            for k, u in enumerate(chunk):
                x_units[u] = x[k]
                eps_units[u] = [e[k] for e in old]

        # ---------------- merge (plms_instance.py:128-135) ------------------------------------------------------
        # The SAME arithmetic at every world size: the N+1 unit latents of every image are laid out in the fixed order
        # [instance][image] and idf_mis_merge reduces them (mean, or crop-and-paste) -- a rank fills in the units it ran and
        # ONE all-gather (RCCL) brings in the others' (below): the merged latent is bit-identical to the 1-rank result.
        # With `image` ownership a rank holds every unit of the images it continues in phase 2: nothing to exchange.
        lat = torch.zeros((n_all, B) + tuple(shape[1:]), device=dev, dtype=torch.float32)
        for (j, b), xv in x_units.items():
            lat[j, b] = xv
        if dist is not None and mode != "image":
            # the ONE hot-path collective: an all-gather of the unit latents each rank OWNS (round 4; rounds 2-3 all-reduced the
            # whole zero-padded stack, (N+1) B latents per rank).  Every rank can enumerate every rank's unit list, so the
            # gathered blocks are scattered into the fixed [instance][image] layout by index -- pure data movement, the stack
            # (and with it the merged latent) is bit-identical to the single-rank one.  Payload per rank: ceil((N+1) B / W)
            # latents of 64 KiB.
            owned = [[(j, b) for j in range(n_all) for b in range(B) if (b + j) % world == r] for r in range(world)]
            cap = max(len(o) for o in owned)
            send = torch.zeros((cap,) + tuple(shape[1:]), device=dev, dtype=torch.float32)
            for k, u in enumerate(owned[rank]):
                send[k] = x_units[u]
            recv = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(recv, send)
            flat = lat.view((n_all * B,) + tuple(shape[1:]))
            for r in range(world):
                if r != rank and owned[r]:
                    idx_r = torch.tensor([j * B + b for (j, b) in owned[r]], device=dev, dtype=torch.long)
                    flat[idx_r] = recv[r][:len(owned[r])]
        if self.crop_and_paste_latents:
            boxes = torch.tensor([[int(v * latent_size) for v in inp["grounding_input"]["boxes"][0][0].tolist()]
                                  for inp in input_all[1:]], dtype=torch.int32, device=dev).reshape(-1, 4)
            merged = ops.mis_merge(lat, boxes, ops.empty(tuple(shape), torch.float32), 1)
        else:
            merged = ops.mis_merge(lat, None, ops.empty(tuple(shape), torch.float32), 0)

        # ---------------- phase 2: shared trajectory, one per image, sharded over images (:138-156) -------------
        out = torch.zeros(tuple(shape), device=dev, dtype=torch.float32)
        if mine:
            idx = torch.tensor(mine, device=dev)
            n = len(mine)
            x = merged[idx].contiguous()
            old = [torch.stack([eps_units[(0, b)][k] for b in mine]) for k in range(len(eps_units[(0, mine[0])]))] \
                if mis_step > 0 else []
            pair = eng.gather_cond(bank, rows([(0, b) for b in mine]))
            for i, step in enumerate(time_range):
                if i < mis_step:
                    continue
                self._apply_alpha(alphas, i)
                index = total - i - 1
                step_next = int(time_range[min(i + 1, len(time_range) - 1)])
                x, e_t = self._plms_step(x, old, index, int(step), step_next, pair, n, guided, guidance_scale)
                self._push(old, e_t)
            out[idx] = x
        if dist is not None:
            # finished images of every rank, again as an all-gather of what each rank owns (images b = r, r + W, ...)
            cap = (B + world - 1) // world
            send = torch.zeros((cap,) + tuple(shape[1:]), device=dev, dtype=torch.float32)
            if mine:
                send[:len(mine)] = out[torch.tensor(mine, device=dev)]
            recv = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(recv, send)
            for r in range(world):
                theirs = [b for b in range(B) if b % world == r]
                if r != rank and theirs:
                    out[torch.tensor(theirs, device=dev)] = recv[r][:len(theirs)]
        input_all[0]["x"] = out
        return out
