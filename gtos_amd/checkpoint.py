"""Checkpoint files in the reference's format.

The reference writes ``torch.save({'args': args, 'model': model.state_dict()}, path)`` (generator/train.py:164) and
reads them back in generator/work.py:70-76,103: ``args`` is the training run's argparse.Namespace (vocabulary file
names, model sizes), ``model`` the state_dict with the parameter names gtos_amd's modules keep.  So a checkpoint written
by either stack loads in the other.
"""
import argparse

import torch

from .vocab import load_vocabs

# constructor arguments of Generator, in order, as attributes of the checkpoint's args (generator/work.py:92-99)
GENERATOR_ARG_NAMES = ('token_char_dim', 'token_dim', 'concept_char_dim', 'concept_dim', 'cnn_filters', 'char2word_dim',
                       'char2concept_dim', 'rel_dim', 'rnn_hidden_size', 'rnn_num_layers', 'embed_dim', 'ff_embed_dim',
                       'num_heads', 'dropout', 'snt_layers', 'graph_layers', 'inference_layers', 'pretrained_file')


def save_checkpoint(path, args, model):
    """args: argparse.Namespace (or dict, stored as a Namespace); model: a gtos_amd or reference Generator."""
    if isinstance(args, dict):
        args = argparse.Namespace(**args)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({'args': args, 'model': sd}, path)


def load_checkpoint(path):
    """-> (args Namespace, state_dict on CPU).  The file is a pickle holding an argparse.Namespace, hence
    weights_only=False: only open checkpoints you trust, exactly as with the reference's torch.load."""
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    return ckpt['args'], ckpt['model']


def generator_args(args):
    a = [getattr(args, n) for n in GENERATOR_ARG_NAMES]
    a[4] = list(a[4]) if not isinstance(a[4], str) else _parse_filters(a[4])
    return a


def _parse_filters(s):
    # train.sh passes --cnn_filters 3 256 (nargs='+', type=int); train.py:102 pairs them up as (width, channels)
    v = [int(x) for x in s.split()]
    return list(zip(v[:-1:2], v[1::2]))


def build_from_checkpoint(path, device, vocabs=None, compute_dtype=torch.float32):
    """Generator with the checkpoint's sizes and weights on ``device`` (eval mode).  ``vocabs`` defaults to the
    vocabulary files named in the checkpoint's args."""
    from .generator import Generator
    args, sd = load_checkpoint(path)
    if vocabs is None:
        vocabs = load_vocabs(args)
    ga = generator_args(args)
    filters = ga[4]
    if filters and not isinstance(filters[0], (tuple, list)):
        ga[4] = list(zip(filters[:-1:2], filters[1::2]))
    ga[-1] = None                                             # pretrained vectors only seed training; the weights are in sd
    # the two flavours differ in the depth-embedding table only: 32 rows (generator.py:38) or 256 (translator/generator.py:39)
    model = Generator(vocabs, *ga, device, depth_size=sd['concept_depth.weight'].shape[0])
    model.load_state_dict(sd)
    model = model.to(device)
    model.set_compute_dtype(compute_dtype)
    model.eval()
    return model, args, vocabs
