import sys
sys.path.insert(0, 'stable-diffusion-on-device_amd'); sys.path.insert(0, 'tools')
import torch
from sdod.amd import ops
from gn_bench import graph_time
d = torch.device('cuda:0')
a = torch.zeros(8, dtype=torch.float16, device=d); b = torch.zeros_like(a)
print('tiny elementwise launch, back to back in a graph: %.2f us per launch' % graph_time(lambda: ops.add(a, b), 50))
x = torch.zeros(2, 8, 8, 1280, dtype=torch.float16, device=d); w = torch.ones(1280, device=d); bb = torch.zeros(1280, device=d)
print('group norm 8x8x1280: %.2f us' % graph_time(lambda: ops.group_norm_nhwc(x, 32, w, bb, 1e-5, True) if hasattr(ops, 'group_norm_nhwc') else ops.add(a, b), 50))
