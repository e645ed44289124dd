"""Initial parameters for the oracle, drawn like the reference draws them.

TEST INFRASTRUCTURE ONLY.  The reference initialises by constructing torch.nn
layers in a fixed order (/root/reference/GAN_functions.py:75-125, :159-176);
the same constructors are called here in the same order purely as RNG
consumers, and the result is returned as a flat dict under the reference's
state_dict keys.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

SEQ_LEN = 16


def _put(sd, prefix, module):
    for k, v in module.state_dict().items():
        sd[f"{prefix}.{k}"] = v.detach().clone()


def generator_state(input_dim, output_dim, vocab_sizes, cfg):
    sd = OrderedDict()
    latent = cfg['latent_dim']
    e_tok = latent // SEQ_LEN
    assert e_tok * SEQ_LEN == latent
    for i, vs in enumerate(vocab_sizes):                          # :85
        _put(sd, f"embeddings.{i}", nn.Embedding(vs, min(100, vs)))
    emb_dim = sum(min(100, vs) for vs in vocab_sizes)
    hidden = cfg['hidden_dims']
    in_dim = input_dim
    for li, h in enumerate(hidden):                               # :89-96
        _put(sd, f"encoder.{3 * li}", nn.Linear(in_dim, h))
        _put(sd, f"encoder.{3 * li + 1}", nn.BatchNorm1d(h))
        in_dim = h
    _put(sd, "mu", nn.Linear(hidden[-1], latent))                 # :97-98
    _put(sd, "log_var", nn.Linear(hidden[-1], latent))
    for bi in range(cfg['residual_blocks']):                      # :101-114
        p = f"residual_attention_blocks.{bi}"
        _put(sd, f"{p}.0.block.0", nn.Linear(e_tok, e_tok))
        _put(sd, f"{p}.0.block.1", nn.BatchNorm1d(e_tok))
        _put(sd, f"{p}.0.block.3", nn.Linear(e_tok, e_tok))
        _put(sd, f"{p}.0.block.4", nn.BatchNorm1d(e_tok))
        _put(sd, f"{p}.1.attention", nn.MultiheadAttention(e_tok, cfg['num_attention_heads'], dropout=0.1,
                                                           batch_first=True))
        if cfg.get('use_sparse_attention', False):                # :43 (gate is created, never used)
            _put(sd, f"{p}.1.gate", nn.Linear(e_tok, 1))
        _put(sd, f"{p}.1.norm", nn.LayerNorm(e_tok))
    in_dim = latent + emb_dim
    for li, h in enumerate(hidden):                               # :117-123
        _put(sd, f"decoder.{3 * li}", nn.Linear(in_dim, h))
        _put(sd, f"decoder.{3 * li + 1}", nn.BatchNorm1d(h))
        in_dim = h
    _put(sd, f"decoder.{3 * len(hidden)}", nn.Linear(in_dim, output_dim))   # :124
    return sd


def discriminator_state(input_dim, vocab_sizes):
    sd = OrderedDict()
    for i, vs in enumerate(vocab_sizes):                          # :161
        _put(sd, f"embeddings.{i}", nn.Embedding(vs, min(100, vs)))
    emb_dim = sum(min(100, vs) for vs in vocab_sizes)
    _put(sd, "model.0", nn.Linear(input_dim + emb_dim, 1024))     # :165-174
    _put(sd, "model.1", nn.BatchNorm1d(1024))
    _put(sd, "model.4", nn.Linear(1024, 512))
    _put(sd, "model.7", nn.Linear(512, 256))
    _put(sd, "model.9", nn.Linear(256, 1))
    return sd


def build_nets(g_mouse, g_human, vocab_sizes, cfg, seed):
    """Four nets in the order of GAN_main.py:125-128 after torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    nets = OrderedDict()
    nets['g_m2h'] = generator_state(g_mouse, g_human, vocab_sizes, cfg)
    nets['g_h2m'] = generator_state(g_human, g_mouse, vocab_sizes, cfg)
    nets['d_m'] = discriminator_state(g_mouse, vocab_sizes)
    nets['d_h'] = discriminator_state(g_human, vocab_sizes)
    return nets


def cwgan_generator_state(x_dim, vocab_sizes, nb_numeric, z_dim):
    """Generator of the conditional WGAN (/root/reference/cWGAN_my.ipynb cell 1): embeddings of width
    min(50, vocab), then Linear-BatchNorm pairs of widths 256, 512, 1024 and the output Linear."""
    sd = OrderedDict()
    for i, vs in enumerate(vocab_sizes):
        _put(sd, f"embeddings.{i}", nn.Embedding(vs, min(50, vs)))
    in_dim = z_dim + sum(min(50, vs) for vs in vocab_sizes) + nb_numeric
    for li, h in enumerate((256, 512, 1024)):
        _put(sd, f"network.{3 * li}", nn.Linear(in_dim, h))
        _put(sd, f"network.{3 * li + 1}", nn.BatchNorm1d(h))
        in_dim = h
    _put(sd, "network.9", nn.Linear(in_dim, x_dim))
    return sd


def cwgan_discriminator_state(x_dim, vocab_sizes, nb_numeric):
    """Critic of the conditional WGAN (cWGAN_my.ipynb cell 2, use_neg_detector=False)."""
    sd = OrderedDict()
    for i, vs in enumerate(vocab_sizes):
        _put(sd, f"embeddings.{i}", nn.Embedding(vs, min(50, vs)))
    in_dim = x_dim + sum(min(50, vs) for vs in vocab_sizes) + nb_numeric
    for li, h in enumerate((1024, 512, 256)):
        _put(sd, f"main_network.{3 * li}", nn.Linear(in_dim, h))
        in_dim = h
    _put(sd, "main_network.9", nn.Linear(in_dim, 1))
    return sd
