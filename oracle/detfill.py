"""Deterministic, machine-independent tensor fill (TEST INFRASTRUCTURE — part of oracle/).

Nothing in the product path (ws-mgmap_amd/) may import this file; it is used by
tools/make_goldens.py (golden capture from the reference), tests/ and
__graft_entry__.smoke() so that weights and inputs never have to be committed
as blobs (SURVEY.md §8c "Deterministic fill").

Formula, for a tensor called `name` with n elements (exact in uint64/float64):

    u_i = ((i * 2654435761 + crc32(name)) mod 2**32) / 2**32 - 0.5        i = 0..n-1

then `* scale` and a cast to float32.  Parameter scales follow a fan-in rule so
that activations stay O(1) through the 17-conv map stack.
"""
import re
import zlib

import numpy as np

_MULT = np.uint64(2654435761)
_MASK = np.uint64(0xFFFFFFFF)


def _u(name: str, n: int) -> np.ndarray:
    c = np.uint64(zlib.crc32(name.encode()))
    i = np.arange(n, dtype=np.uint64)
    h = (i * _MULT + c) & _MASK
    return h.astype(np.float64) / 4294967296.0 - 0.5


def uniform(name: str, shape, scale: float = 1.0) -> np.ndarray:
    """float32 array of `shape`, values in [-0.5, 0.5) * scale."""
    n = int(np.prod(shape)) if len(shape) else 1
    return (_u(name, n) * scale).astype(np.float32).reshape(shape)


def positive(name: str, shape, lo: float = 0.5) -> np.ndarray:
    """|u| + lo  (BatchNorm weight / running_var)."""
    n = int(np.prod(shape)) if len(shape) else 1
    return (np.abs(_u(name, n)) + lo).astype(np.float32).reshape(shape)


def tokens(name: str, batch: int, n_tok, max_len: int = 200, vocab: int = 2504) -> np.ndarray:
    """int64 [batch, max_len]: row b has n_tok[b] tokens in 1..vocab-1, then zeros
    (reference pads instructions with 0 to 200: config/default.py:83)."""
    if np.isscalar(n_tok):
        n_tok = [int(n_tok)] * batch
    out = np.zeros((batch, max_len), dtype=np.int64)
    c = np.uint64(zlib.crc32(name.encode()))
    for b in range(batch):
        i = np.arange(n_tok[b], dtype=np.uint64) + np.uint64(b * max_len)
        h = (i * _MULT + c) & _MASK
        out[b, : n_tok[b]] = 1 + (h % np.uint64(vocab - 1)).astype(np.int64)
    return out


_ALIAS = [
    # MapDecoder / ResNetUNet register resnet18 children twice (map_encoder.py:75-81,
    # unet_encoder.py:34-47): `layer0 = Sequential(conv1, bn1, relu)`,
    # `layer1 = Sequential(maxpool, layer1)`, `layer2..4 = base_layers[5..7]`.
    (re.compile(r"^(.*(?:map_decoder|rgb_encoder\.base_model))\.layer0\.0\.(.*)$"), r"\1.base_model.conv1.\2"),
    (re.compile(r"^(.*(?:map_decoder|rgb_encoder\.base_model))\.layer0\.1\.(.*)$"), r"\1.base_model.bn1.\2"),
    (re.compile(r"^(.*(?:map_decoder|rgb_encoder\.base_model))\.layer1\.1\.(.*)$"), r"\1.base_model.layer1.\2"),
    (re.compile(r"^(.*rgb_encoder\.base_model)\.layer([234])\.(.*)$"), r"\1.base_model.layer\2.\3"),
]


def canon(key: str) -> str:
    """Canonical name of a state_dict key (aliases of one tensor map to one name)."""
    for pat, rep in _ALIAS:
        if pat.match(key):
            return pat.sub(rep, key)
    return key


def state_value(key: str, shape) -> np.ndarray:
    """Value for a state_dict entry `key` (reference key names, e.g.
    'net.map_encoder.cnn.0.weight')."""
    key = canon(key)
    v = _state_value(key, shape)
    # the frozen RGB encoder eats raw 0..255 pixels (unet_encoder.py:68, no normalisation):
    # shrink the two convs that see the image so synthetic activations stay O(1).
    if "rgb_encoder" in key and key.endswith(("conv_original_size0.0.weight", "base_model.base_model.conv1.weight")):
        v = v / np.float32(128.0)
    return v


def _state_value(key: str, shape) -> np.ndarray:
    leaf = key.rsplit(".", 1)[-1]
    shape = tuple(int(s) for s in shape)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_var":
        return positive(key, shape)
    if leaf == "running_mean":
        return uniform(key, shape, 0.2)
    if leaf == "_scale":  # MGMapNet attention scale buffer (mg_map_policy.py:132) — keep its real value
        return np.asarray(1.0 / (256 ** 0.5), dtype=np.float32).reshape(shape)
    is_norm = len(shape) == 1 and leaf == "weight"
    if is_norm:  # BatchNorm gamma
        return positive(key, shape)
    if leaf.startswith("bias") or leaf == "_bias" or len(shape) <= 1:
        return uniform(key, shape, 0.2)
    # weights: [out, in, *k] (conv / linear / rnn) or ConvTranspose [in, out, *k]; embeddings [rows, dim]
    if "embedding" in key:
        return uniform(key, shape, 2.0)
    fan_in = int(np.prod(shape[1:]))
    if "map_classfier.0." in key:  # ConvTranspose2d(64,32,4,s2): each output sees in*k*k/s^2 taps
        fan_in = shape[0] * shape[2] * shape[3] // 4
    return uniform(key, shape, float(np.sqrt(24.0 / max(fan_in, 1))))


def fill_state_dict(shapes: dict) -> dict:
    """{key: shape} -> {key: ndarray}."""
    return {k: state_value(k, s) for k, s in shapes.items()}
