import sys, torch
sys.path.insert(0, '.')
import vfa_amd
from vfa_amd import ops, vfa_op
from vfa_amd.synthetic import make_workload
dev = torch.device('cuda:0')
for name in ('multiviewc_156x156x5', 'multiviewc_200x200x1'):
    wl = make_workload(name, channels=256, seed=3, n_cam=3)
    grid = wl['grid'][:, 60:60+16, 40:40+24].contiguous().to(dev)
    torch.manual_seed(1)
    mods = [vfa_amd.VFA(256, grid_height=wl['grid_height'], cube_size=wl['cube_size'], args=wl['args']).to(dev) for _ in range(3)]
    lats = [torch.cat([wl['features'][c][s] for c in range(3)]).to(dev) for s in range(3)]
    calibs = wl['calibs'].to(dev)
    lats[0][0, 3, 0, 0] = float('nan')
    with torch.no_grad():
        ii = ops.integral_images(lats)
        print(name, 'integral nan frac ch3 cam0', torch.isnan(ii[0][0, :, :, 3]).float().mean().item(), 'absmax', [hex(a.max().item()) for a in ii.absmax])
        for t in (2, 3, 6):
            out = vfa_op.pipe_frame(mods, lats, calibs, grid, terms=t)
            print(' pipe terms', t, 'nan frac', torch.isnan(out).float().mean().item())
        if name.endswith('x1'):
            for t in (2, 3):
                vfa_op.COLLAPSE_TERMS = t
                out = vfa_op.fused_frame(mods, lats, calibs, grid)
                print(' serial terms', t, 'nan frac', torch.isnan(out).float().mean().item())
