#!/usr/bin/env python3
"""Attainable bound of the coupling-layer forward kernel (16 x 16 tiles, this algorithm) from the fp64 work it
cannot avoid -- written to profiles/attainable.json and carried in bench.py's roofline.

fp64 MFMA and fp64 VALU share one DP pipe per SIMD on gfx950 (profiles/r01_microbench_fp64.txt: 77 TF MFMA, 68 TF FMA,
35 + 35 TF together; SQ_VALU_MFMA_COEXEC_CYCLES = 0 in profiles/r02_pmc_kernels_fullbatch.json), every VALU
instruction of a wave occupies it for ~4.2 cycles (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU), a v_mfma_f64_16x16x4 for 64."""
import json, os
L, B, TILE = 64, 128, 16
mfma = 36 + 192                                  # conv1 on tile+2, frozen taps only (12 tiles x 3) + conv2 on the live lines of tile+1 (8 x 24)
sig = (20 * 20 + 14 * 18) * 8                    # sigmoids: h1 on tile+2, h2 on the live lines of tile+1, 8 channels
work = {                                         # SIMD-cycles per workgroup
    'mfma_228_x_64': mfma * 64,
    'sigmoid_5216_x_26_ops': sig * 26 * 4 / 64,
    'sincos_484_x_45_ops': 22 * 22 * 45 * 4 / 64,
    'conv3_active_sites_fma': 64 * 8 * 9 * 3 * 4 / 64,
    'tan_mixture_transform_and_logJ': 64 * 2 * 120 * 4 / 64,
}
cyc_wg = sum(work.values()) / 4                  # four SIMDs per CU
wgs_per_cu = B * (L // TILE) ** 2 / 256
flops = 1872 * L * L * B                         # dense accounting, SURVEY 8d
out = {'kernel': 'k_flow_fwd<16,16>', 'assumptions': __doc__.split('\n\n')[1].replace('\n', ' '),
       'simd_cycles_per_workgroup': {k: round(v) for k, v in work.items()}, 'cycles_per_workgroup_per_cu': round(cyc_wg),
       'halo_factors': {'conv1': 400 / 256, 'conv2_live': 252 / 256, 'sincos': 484 / 256}}
for name, ghz in (('at_2.4GHz_nominal', 2.4), ('at_2.1GHz_held_under_this_load', 2.1)):
    t = wgs_per_cu * cyc_wg / (ghz * 1e9)
    out[name] = {'full_batch_launch_us': round(t * 1e6, 2), 'TFLOPs_dense_accounting': round(flops / t / 1e12, 2),
                 'frac_of_fp64_peak': round(flops / t / 1e12 / 78.6, 4)}
out['note'] = ('bound of THIS algorithm at this tile size (halo recompute, sigmoid at fp64 accuracy); the 2.1 GHz figure is the clock '
               'implied by in-kernel cycle stamps against wall time (tools/lifetime.py), the chip lowers it under fp64 load')
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'attainable.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
