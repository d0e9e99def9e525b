import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/stable-diffusion-on-device_amd')
from sdod.amd import ops
d = torch.device('cuda:0')
def conv_ref(x, w, b):
    cout = w.shape[0]
    return F.conv2d(x.float().permute(0, 3, 1, 2), w.float().reshape(cout, 3, 3, -1).permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
def run(tile, n, h, w, cin, cout, uniform, label):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(n, h, w, cin, generator=g)).half()
    q = torch.randint(0, 256, (cout, 9 * cin), generator=g, dtype=torch.uint8)
    if uniform:
        scale = torch.full((cout,), 0.01); offset = torch.full((cout,), -128.0)
    else:
        scale = (torch.rand(cout, generator=g) * 0.5 + 0.75) * 2.0 / 255 * (9 * cin) ** -0.5 * torch.where(torch.arange(cout) < cout // 2, 1.0, 1.7); offset = torch.where(torch.arange(cout) < cout // 2, torch.tensor(-128.0), torch.tensor(-77.0))
    wf = (q.float() + offset[:, None]) * scale[:, None]
    bias = torch.randn(cout, generator=g)
    ref = conv_ref(x, wf, bias)
    try:
        out = ops.gemm(x.to(d), q.to(d), bias.to(d), conv=dict(stride=1), w_scale=scale.to(d), w_off=(offset + 128).to(d), tile=tile, split_k=0, out=torch.full((n, h, w, cout), float('nan'), dtype=torch.float16, device=d)).float().cpu()
    except Exception as ex:
        print(label, 'declined', str(ex)[:60]); return
    err = (out - ref)
    rel = float(err.norm() / ref.norm())
    print(f'{label} tile{tile} {n}x{h}x{w} {cin}->{cout}: rel {rel:.3e}')
    if rel > 2e-3:
        percol = (err.reshape(-1, cout).norm(dim=0) / ref.reshape(-1, cout).norm(dim=0))
        print('  bias-only error per 16 columns:', ' '.join(f'{float(((out - ref).reshape(-1, cout).mean(0) / bias)[i:i+16].mean()):.2f}' for i in range(0, cout, 16)))
        print('  per 16 columns:', ' '.join(f'{float(percol[i:i+16].mean()):.2f}' for i in range(0, cout, 16)))
        perrow = (err.reshape(-1, cout).norm(dim=1) / ref.reshape(-1, cout).norm(dim=1))
        print('  per 16 rows (first 16 groups):', ' '.join(f'{float(perrow[i:i+8].mean()):.2f}' for i in range(0, min(256, perrow.numel()), 8)))
        # ratio out/ref
        ratio = (out.reshape(-1, cout) * ref.reshape(-1, cout)).sum(0) / (ref.reshape(-1, cout) ** 2).sum(0)
        print('  out/ref projection per 16 columns:', ' '.join(f'{float(ratio[i:i+16].mean()):.2f}' for i in range(0, cout, 16)))
        # fp16 kernel sanity with dequantised weights
        o2 = ops.gemm(x.to(d), wf.half().to(d), bias.to(d), conv=dict(stride=1), tile=tile, split_k=1).float().cpu()
        print('  fp16 weights same tile: rel', float((o2 - ref).norm() / ref.norm()))
for tile in (51, 52, 38):
    run(tile, 3, 8, 8, 128, 80, False, 'images8')
    run(tile, 3, 8, 8, 64, 80, False, 'images8 1chunk')
    run(tile, 3, 8, 8, 64, 80, True, 'images8 1chunk uniform')
    run(tile, 3, 8, 8, 128, 64, False, 'images8 n64')
    run(tile, 2, 24, 24, 128, 80, False, 'rows24')
