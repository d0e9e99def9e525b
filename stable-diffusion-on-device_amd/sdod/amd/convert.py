"""Real-checkpoint loader: a CompVis Stable Diffusion v1.x checkpoint (`sd-v1-4.ckpt` or its
`.safetensors` twin) -> the four weight containers libsdod_setup / Txt2Img(models_dir=...) read.

The reference never touches a checkpoint at run time: its models arrive as serialized QNN graphs made offline by
`todlc.py` from ONNX exports of the ldm model (README.md:60-65, context.cpp:105-115).  This module is the same offline
step for the MI355X build.  It runs on a host without a GPU: the engine's parameter tables are host-side metadata.

    python -m sdod.amd.convert --ckpt sd-v1-4.ckpt --out models/          # + --tokenizer-vocab bpe_simple_vocab_16e6.txt.gz

Only loaders that execute nothing from the file are used: safetensors, or torch.load(weights_only=True).
"""
import argparse
import os

import torch

from . import engine as E
from . import weights

# graph -> (prefix inside an ldm checkpoint, container file stem in models_dir)
GRAPHS = {
    'unet': ('model.diffusion_model.', 'unet'),
    'temb': ('model.diffusion_model.', 'temb'),          # time MLP + every ResBlock's emb_layers projection (TEMB graph)
    'vae': ('first_stage_model.', 'vae_decoder'),
    'text': ('cond_stage_model.transformer.', 'text_encoder'),
}


def read_checkpoint(path):
    """-> flat {name: tensor}.  `.safetensors` through safetensors; anything else through torch.load(weights_only=True)
    (which refuses pickled code); a top-level 'state_dict' entry (ldm / lightning checkpoints) is unwrapped."""
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        return load_file(path, device='cpu')
    sd = torch.load(path, map_location='cpu', weights_only=True, mmap=True)
    if isinstance(sd, dict) and 'state_dict' in sd and isinstance(sd['state_dict'], dict):
        sd = sd['state_dict']
    return sd


# SD v2.x (public ldm v2 checkpoints): same UNet / VAE prefixes, the text tower is open_clip's model under `.model.`
GRAPHS_SD21 = dict(GRAPHS, text=('cond_stage_model.model.', 'text_encoder'))


def parameter_tables(cfg=None):
    """{graph: [(name, shape), ...]} of the graphs for `cfg` (default SD v1.x; E.sd21_config() for SD v2.1); no device needed"""
    cfg = cfg or E.sd14_config()
    return {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(),
            'vae': E.VaeDecoder(cfg, 1).param_table(), 'text': E.TextEncoder(cfg, 1).param_table()}


def split_state_dict(sd, tables=None, dtype=torch.float16):
    """Pick every graph's parameters out of a full checkpoint state dict.  Raises KeyError naming what is missing and
    ValueError on a shape mismatch; entries the graphs do not use (EMA copies, the VAE encoder, position_ids,
    loss / scheduler buffers) are ignored.  Returns ({graph: {name: tensor}}, [unused keys])."""
    tables = tables or parameter_tables()
    names = {n for n, _ in tables.get('text', [])}
    prefixes = GRAPHS_SD21 if 'ln_final.weight' in names else GRAPHS      # open_clip text tower => an SD2.x checkpoint
    out, used, missing = {}, set(), []
    for graph, table in tables.items():
        prefix = prefixes[graph][0]
        part = {}
        for name, shape in table:
            key = prefix + name
            if key not in sd and graph == 'text' and name.startswith('text_model.') and prefix + name[len('text_model.'):] in sd:
                key = prefix + name[len('text_model.'):]          # transformers >= 5 drops the text_model. level
            if key not in sd:
                missing.append(key)
                continue
            t = sd[key]
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f'{key}: checkpoint shape {tuple(t.shape)} != graph shape {tuple(shape)}')
            part[name] = t.to(dtype) if t.is_floating_point() else t.float().to(dtype)
            used.add(key)
        out[graph] = part
    if missing:
        raise KeyError(f'{len(missing)} parameter(s) missing from the checkpoint, e.g. {missing[:5]}')
    return out, sorted(k for k in sd if k not in used)


def convert(ckpt_path, out_dir, dtype=torch.float16, cfg=None):
    """checkpoint file -> out_dir/{unet,temb,vae_decoder,text_encoder}.sdodw; returns the list of files written"""
    parts, _ = split_state_dict(read_checkpoint(ckpt_path), parameter_tables(cfg), dtype)
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for graph, part in parts.items():
        path = os.path.join(out_dir, GRAPHS[graph][1] + '.sdodw')
        weights.save(path, part)
        written.append(path)
    return written


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--ckpt', required=True, help='sd-v1-x .ckpt / .safetensors (ldm key names)')
    ap.add_argument('--out', required=True, help='models_dir to write')
    ap.add_argument('--fp32', action='store_true', help='keep fp32 payloads (the engine converts at load)')
    ap.add_argument('--model', default='sd14', choices=['sd14', 'sd21'], help='sd21: SD v2.x shapes and open_clip text-tower key names')
    ap.add_argument('--tokenizer-vocab', help='bpe_simple_vocab_16e6.txt.gz, or a directory with HF vocab.json + merges.txt: '
                                              'also write ctokenizer.txt')
    a = ap.parse_args(argv)
    cfg = E.sd21_config() if a.model == 'sd21' else None
    for p in convert(a.ckpt, a.out, torch.float32 if a.fp32 else torch.float16, cfg):
        print('wrote', p)
    if a.tokenizer_vocab:
        from . import tokenizer_file
        print('wrote', tokenizer_file.generate(a.tokenizer_vocab, os.path.join(a.out, 'ctokenizer.txt')))


if __name__ == '__main__':
    main()
