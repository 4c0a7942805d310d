"""PREDICTED 1 / 2 / 4 / 8-GPU throughput of the Multi-instance Sampler bench -- a MODEL, not a measurement:
this container's GPU boxes have one GPU.  Inputs: measured graph-replay times of full-model UNet forwards per row width on ONE
MI355X (tools/profile_forward.py <rows> 20 graph) and the forward schedule `host/samplers.PLMSSamplerInst` forms per rank:

  phase 1 (ms + 1 = 19 evaluations): the rank's (instance, image) units in chunks of `max_units` = 64 -> 128-row forwards plus one
           remainder forward of 2 x (units mod 64) rows;
  phase 2 (S - ms = 32 evaluations): the rank's images, 2 rows each;
  collectives: ONE all-gather of the owned unit latents (64 KiB each) at the merge + ONE of the finished images -- priced at
           xGMI ring rates (7 links x ~153 GB/s per GPU, MI355X_MICROARCH.md / the task's hardware notes; 50 GB/s effective per
           rank assumed, i.e. far below link rate, plus 50 us per collective) -- negligible against the step either way;
  `instance` ownership overhead: every rank evaluates the hoisted unconditional row of the FIRST evaluation for every image it
           holds a unit of (weak scaling at W = 8: 256 uncond + 288 cond rows instead of 32 + 288).

Usage: python tools/scaling_model.py '{"2": 10.4, "4": .., "8": .., "16": .., "18": .., "36": .., "64": .., "72": .., "128": ..}'
Row widths that were not measured are interpolated linearly between the neighbouring measured widths."""
import json
import sys

N_INST, S, MIS, MAX_UNITS = 8, 50, 0.36, 64       # (the sampler's default is 128 since the end of round 6: the committed prediction
                                                    #  tables were made with 64; for 128 add a measured "256" width to the input)


def interp(tab, rows):
    ks = sorted(tab)
    if rows in tab:
        return tab[rows]
    if rows <= ks[0]:
        return tab[ks[0]]
    if rows >= ks[-1]:
        return tab[ks[-1]] * rows / ks[-1]
    lo = max(k for k in ks if k < rows)
    hi = min(k for k in ks if k > rows)
    return tab[lo] + (tab[hi] - tab[lo]) * (rows - lo) / (hi - lo)


def step_ms(tab, world, images_total):
    ms = int(S * MIS)
    units = (N_INST + 1) * images_total
    u_rank = -(-units // world)                               # the busiest rank
    i_rank = -(-images_total // world)
    full, rem = divmod(u_rank, MAX_UNITS)
    eval_ms = full * interp(tab, 2 * MAX_UNITS) + (interp(tab, 2 * rem) if rem else 0.0)
    p1 = (ms + 1) * eval_ms
    p2 = (S - ms) * interp(tab, 2 * i_rank)
    # first evaluation, `instance` ownership at W > 1: the unconditional row of every image the rank holds a unit of
    extra_rows = 0 if world == 1 else min(images_total, u_rank) - i_rank
    first = extra_rows * interp(tab, 128) / 128.0
    comm = 0.0 if world == 1 else 2 * 0.05 + (units + images_total) * 65536 / 50e9 * 1e3
    return p1 + p2 + first + comm, dict(phase1_ms=round(p1, 1), phase2_ms=round(p2, 1), first_eval_extra_ms=round(first, 1),
                                        collectives_ms=round(comm, 2), phase1_rows=2 * min(u_rank, MAX_UNITS), phase2_rows=2 * i_rank)


def main():
    tab = {int(k): float(v) for k, v in json.loads(sys.argv[1]).items()}
    out = {"model": "predicted from single-GPU graph-replay times per forward width; NOT measured on multi-GPU hardware",
           "forward_ms_by_rows": tab, "weak_32_images_per_gpu": {}, "strong_8_images_total": {}}
    base_w = base_s = None
    for w in (1, 2, 4, 8):
        t, d = step_ms(tab, w, 32 * w)
        v = 32 * w / (t * 1e-3)
        base_w = base_w or v
        out["weak_32_images_per_gpu"][str(w)] = dict(img_per_s=round(v, 2), step_s=round(t * 1e-3, 2), efficiency=round(v / (w * base_w), 3), **d)
        t, d = step_ms(tab, w, 8)
        v = 8 / (t * 1e-3)
        base_s = base_s or v
        out["strong_8_images_total"][str(w)] = dict(img_per_s=round(v, 2), step_s=round(t * 1e-3, 3), speedup=round(v / base_s, 2),
                                                    efficiency=round(v / (w * base_s), 3), **d)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
