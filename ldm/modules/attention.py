from instancediffusion_amd.host.attention import (  # noqa: F401
    BasicTransformerBlock, CrossAttention, FeedForward, GatedSelfAttentionDense, SelfAttention, SpatialTransformer)
