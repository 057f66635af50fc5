#!/usr/bin/env python3
"""Fold rocprofv3 --pmc per-dispatch CSVs (tools/collect_pmc.sh) into mean-per-launch values per kernel."""
import csv, glob, json, os, sys
from collections import defaultdict

FAMILIES = [('k_ft_small', 'k_ft_small (small-lattice fused trajectory / force / action)'),
            ('k_flow_bwd_gather', 'k_flow_bwd_gather<16,16> (coupling-layer backward wrt x, force path)'),
            ('k_flow_fwd', 'k_flow_fwd<16,16> (coupling-layer forward)'),
            ('k_leap_rows', 'k_leap_rows<8> (fused plain-HMC leapfrog step, row strips, 16-byte accesses)'),
            ('k_force<1', 'k_force<1> (fused plain-HMC leapfrog step, 16 x 16 tiles)'),
            ('k_hmc_trajectory', 'k_hmc_trajectory (single-launch plain-HMC trajectory)')]


def family(name):
    for key, label in FAMILIES:
        if key in name:
            return label
    return None


LAUNCH_SHAPE = ('bench.py config 3 with two chain groups: one launch of a coupling-layer kernel = one layer over ONE chain group = '
                '64 chains x 16 tiles = 1024 workgroups of 16x16 sites (L=64, fp64); k_force<1> / k_hmc_trajectory: all 128 chains')
COMMIT = None


def csrc_sha16():
    """fingerprint of the sources the library that was PROFILED was built from (FTHMC_LIB or the in-tree one), as the library
    reports it; falls back to the sources on disk for a library without one"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.lib_sha16() or bench.csrc_sha16()


def main(src, dst):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                fam = family(row.get('Kernel_Name', ''))
                if fam is None:
                    continue
                a = acc[fam][row['Counter_Name']]
                a[0] += float(row['Counter_Value']); a[1] += 1
    out = {
        'command': 'rocprofv3 --pmc <group> --output-format csv -- python3 <bench.py --steps 2 --warmup 1 --no-cpu-baseline | tools/kernel_loop.py>'
                   '  (one pass per counter group: tools/collect_pmc.sh / tools/pmc_kernels.sh)',
        'units': 'FETCH_SIZE / WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide '
                 'streaming reads, MI355X_MICROARCH.md HBM section; our reads are 8 B/lane, uncalibrated: raw values kept); '
                 'SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_VALU in quad-cycles summed over waves or SIMDs; '
                 'SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; SQ_INSTS_* wave-instructions; '
                 'GRBM_GUI_ACTIVE summed over 8 XCDs',
        'launch_shape': LAUNCH_SHAPE,
        'commit': COMMIT,
        'csrc_sha16': csrc_sha16(),           # fingerprint of fthmc_amd/csrc at collection time (bench.py reports `stale` against it)
        'kernels': {fam: {c: {'launches': v[1], 'mean_per_launch': v[0] / v[1]} for c, v in sorted(cs.items())}
                    for fam, cs in acc.items()},
    }
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1)
    for fam, cs in out['kernels'].items():
        print(fam)
        for c, v in cs.items():
            print(f'   {c:34s} {v["mean_per_launch"]:.4g}  ({v["launches"]} launches)')


if __name__ == '__main__':
    if len(sys.argv) > 3 and sys.argv[3]:
        LAUNCH_SHAPE = sys.argv[3]
    if len(sys.argv) > 4:
        COMMIT = sys.argv[4]
    main(sys.argv[1], sys.argv[2])
