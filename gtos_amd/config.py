"""Model construction for the BASELINE.json configurations (defaults of /root/reference/generator/train.sh:10-26)."""
from collections import namedtuple

import torch

from . import synth

VocabSpec = namedtuple("VocabSpec", ["size", "padding_idx"])


def default_vocabs(vocab=None):
    v = dict(synth.DEFAULT_VOCAB, **(vocab or {}))
    return {k: VocabSpec(s, 0) for k, s in v.items()}


def generator_args(cfg):
    """Positional args of Generator(...) after ``vocabs`` (train.sh: char dims 32, word/concept dim 300, cnn 3x256,
    char2word/concept 128, rel_dim 100, GRU 256x2, snt_layers 1, inference_layers 3, dropout 0.2)."""
    return dict(word_char_dim=32, word_dim=300, concept_char_dim=32, concept_dim=300, cnn_filters=[(3, 256)],
                char2word_dim=128, char2concept_dim=128, rel_dim=100, rnn_hidden_size=256, rnn_num_layers=2,
                embed_dim=cfg["d"], ff_embed_dim=cfg["ff"], num_heads=cfg["H"], dropout=0.2, snt_layers=1,
                graph_layers=cfg["layers"], inference_layers=3, pretrained_file=None)


def build_generator(cls, cfg_name, device, seed=19940117, dropout=None, depth_size=None, **extra):
    cfg = synth.CONFIGS[cfg_name]
    args = generator_args(cfg)
    if dropout is not None:
        args["dropout"] = dropout
    torch.manual_seed(seed)                    # identical initialisation on every rank (train.py:98-100)
    depth = depth_size or (256 if cfg["kind"] == "dep" else 32)
    return cls(default_vocabs(), device=device, depth_size=depth, **args, **extra)
