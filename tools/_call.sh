L=instancediffusion_amd
for r in 1 2; do timeout 200 tools/ubench/attn_harness $L/libidf_attn_v0.so 128 1 $L/libidf_attn_v1.so $L/libidf_attn_v4a1.so $L/libidf_attn_v4a2.so $L/libidf_attn_v4a3.so 2>&1 | grep -E "d=40" ; done
