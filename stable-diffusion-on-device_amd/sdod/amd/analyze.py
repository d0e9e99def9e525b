"""Per-launch latency views of an engine graph: the two tables the reference's analyze_results.py prints
for its HTP per-layer profiles -- the ten slowest layers (analyze_results.py:127-131) and latency by operator type with
its share of the total (:133-146) -- fed from HIP-event timings of every launch of a graph (Graph.profile()).

    python -m sdod.amd.analyze unet --hw 64 --op-summary
"""
import argparse
import re


def op_type(label, detail=''):
    """operator class of one launch, from the engine's launch label + shape string"""
    if label.startswith('gemm'):
        if detail.startswith('conv3'):
            return 'Conv 3x3 (implicit GEMM)'
        if detail.startswith('conv1'):
            return 'Conv 1x1'
        return 'Linear / MatMul'
    if label.startswith('attn'):
        return 'Attention (QK^T softmax PV)'
    if label.startswith(('group_norm', 'gn_')):
        return 'GroupNorm(+SiLU)'
    if label.startswith('layer_norm'):
        return 'LayerNorm'
    if label.startswith('softmax'):
        return 'Softmax'
    if label.startswith('splitk_reduce'):
        return 'Split-K reduce + epilogue'
    if label.startswith(('geglu', 'act', 'add', 'silu')):
        return 'Elementwise'
    if label.startswith(('nchw', 'nhwc', 'latent', 'im2col', 'concat', 'embedding', 'timestep')):
        return 'Layout / gather'
    return 'Other'


def summarize(ms, table, details, top=10):
    """ms: per-launch milliseconds; table: [(label, flops, bytes)]; details: [shape string].
    -> (top launches [(name, us)], by type [(type, us, percent, launches)], total us)"""
    rows = [(f'{i:03d} {lab} {det}'.strip(), 1e3 * t) for i, (t, (lab, _, _), det) in enumerate(zip(ms, table, details))]
    total = sum(us for _, us in rows)
    slow = sorted(rows, key=lambda r: -r[1])[:top]
    agg = {}
    for (t, (lab, _, _), det) in zip(ms, table, details):
        e = agg.setdefault(op_type(lab, det), [0.0, 0])
        e[0] += 1e3 * t
        e[1] += 1
    by_type = sorted(((k, us, round(100.0 * us / total, 2) if total else 0.0, n) for k, (us, n) in agg.items()), key=lambda r: -r[1])
    return slow, by_type, total


def format_table(rows, headers):
    try:
        from tabulate import tabulate
        return tabulate(rows, headers=headers, floatfmt='.1f')
    except ImportError:
        widths = [max(len(str(h)), *(len(f'{r[i]:.1f}' if isinstance(r[i], float) else str(r[i])) for r in rows)) for i, h in enumerate(headers)]
        fmt = lambda r: '  '.join((f'{c:.1f}' if isinstance(c, float) else str(c)).ljust(w) for c, w in zip(r, widths))
        return '\n'.join([fmt(headers), fmt(['-' * w for w in widths])] + [fmt(r) for r in rows])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('graph', choices=['unet', 'vae', 'text'])
    ap.add_argument('--hw', type=int, default=64, help='latent height = width')
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--op-summary', action='store_true')
    ap.add_argument('--regex', help='only launches whose name matches')
    a = ap.parse_args(argv)
    import torch
    from . import engine as E, weights as Wt
    cfg = E.sd14_config(a.hw, a.hw)
    g = {'unet': E.UNet, 'vae': E.VaeDecoder, 'text': E.TextEncoder}[a.graph](cfg, a.batch if a.graph != 'vae' else 1)
    g.load_state_dict(Wt.synthetic_state_dict(g.param_table(), seed=1, dtype=torch.float16))
    g.finalize()
    g.execute()
    ms, table, details = g.profile(iters=5), g.op_table(), g.op_details()
    if a.regex:
        rx = re.compile(a.regex)
        keep = [i for i, ((lab, _, _), det) in enumerate(zip(table, details)) if rx.search(f'{lab} {det}')]
        ms, table, details = [ms[i] for i in keep], [table[i] for i in keep], [details[i] for i in keep]
    slow, by_type, total = summarize(ms, table, details)
    print(a.graph)
    print(format_table(slow, ['Launch', 'Latency (us)']))
    print()
    if a.op_summary:
        print(format_table(by_type, ['Op.', 'Latency (us)', '% Latency', 'Launches']))
    print('Total latency of launches (ms):', round(total / 1e3, 4))


if __name__ == '__main__':
    main()
