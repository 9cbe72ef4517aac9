import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from parity_cases import sweep_case
from util import run_case
from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer
from eogs2_amd.rasterizer import last_exact_token
dev = torch.device('cuda:0')
for seed in [int(x) for x in sys.argv[1:]]:
    case, name = sweep_case(seed)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    tok = int(last_exact_token(dev))
    P = case['means3D'].shape[0]; H, W = int(case['H']), int(case['W']) if 'H' in case else (0, 0)
    nt = ((W + 7) // 8) * ((H + 7) // 8)
    slots = tok & 0x7FFFFFFF
    print(seed, name, 'P', P, 'HxW', H, W, 'pairs', slots, 'frac of tiles per Gaussian %.4f' % (slots / max(P * nt, 1)), 'btf', (tok >> 60) & 1)
