import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from fthmc_amd import ops
from oracle import ref_cpu as R
L = 64
gen = torch.Generator().manual_seed(1331)
w = ops.pack_weights(R.default_flow(1, gen), device='cuda')
names = {'flow_fwd': ['', 'plaq+sincos', 'conv1', 'conv2', 'conv3', 'transform1', 'finish+store'],
         'flow_bwd': ['', 'load+xform', 'conv3T', 'conv2T', 'conv1T', 'store'],
         'flow_bwd_train': ['', 'load+xform', 'conv3T', 'conv2T', 'conv1T', 'store'],     # backward with the pre-activation gradients written (training)
         'flow_wgrad': ['', 'issue loads', 'fill LDS', 'conv2/conv1 MFMA', 'barrier', 'h2 fill', 'conv3 VALU + stores']}
for B in [int(a) for a in sys.argv[1:]] or (16, 32, 48, 64, 96, 128, 256):
    x = ((torch.rand(B, 2, L, L, generator=gen, dtype=torch.float64) * 2 - 1) * math.pi).cuda()
    for kind in ('flow_fwd', 'flow_bwd', 'flow_bwd_train', 'flow_wgrad'):
        cyc = ops.profile_stages(kind, x, w, mu=0, off=1, beta=6.0)
        tot = sum(cyc[:7]) if kind in ('flow_fwd', 'flow_wgrad') else sum(cyc[:6])
        ms = ops.time_kernel(kind, x, w, mu=0, off=1, beta=6.0, reps=30) if kind in ('flow_fwd', 'flow_bwd') else float('nan')
        if kind == 'flow_fwd' and any(cyc[7:13]): print('      inside conv1 (tile 0: mfma, epilogue; tile 1: mfma, epilogue), conv2 (mfma, epilogue):', ' '.join(f'{c:.0f}' for c in cyc[7:13]))
        if kind == 'flow_bwd' and 'diag' in os.environ.get('FTHMC_LIB', '') and any(cyc[6:10]):
            t2 = sum(cyc[1:3])
            print('      inside conv2T, cycles after the conv3T barrier: wave 0 tile 0 mfma / epilogue, tile 1 mfma / epilogue; wave 4 mfma / epilogue:', ' '.join(f'{c - t2:.0f}' for c in cyc[6:12]))
        if kind in ('flow_fwd', 'flow_bwd') and cyc[15] > 0:
            print(f'      {kind} shader clock while this launch runs: {tot / (cyc[15] * 10e-9) / 1e9:.3f} GHz ({tot:.0f} cycles in {cyc[15] * 10:.0f} ns per workgroup)')
        if kind == 'flow_wgrad': print(f'      flow_wgrad workgroup: prologue {cyc[9]:.0f}, the walk {cyc[10]:.0f}, epilogue {cyc[11]:.0f} cycles (stages below: its last item)')
        print(f'B={B:3d} WGs={B*16:5d} {kind}: {ms*1e3:7.2f} us; lifetime {tot:6.0f} cycles/WG; ' + ', '.join(f'{n} {c:.0f}' for n, c in zip(names[kind][1:], cyc[1:]) if n), flush=True)
