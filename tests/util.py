"""Shared helpers for tests: golden loading, deterministic parameter dicts."""
import hashlib
import json
import os

import numpy as np
import torch

from oracle import detfill

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def state_spec():
    with open(os.path.join(GOLDEN, "state_dict_spec.json")) as f:
        return json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_params(grad=True, device="cpu"):
    """{reference state_dict key: tensor}; aliased keys share one tensor."""
    P, cache = {}, {}
    for k, v in state_spec().items():
        ck = detfill.canon(k)
        if ck not in cache:
            t = T(detfill.state_value(k, v["shape"]))
            if v["dtype"] == "int64":
                t = t.long()
            t = t.to(device)
            if grad and v["trainable"]:
                t.requires_grad_(True)
            cache[ck] = t
        P[k] = cache[ck]
    return P


def state_dict_values():
    """{key: tensor} suitable for module.load_state_dict (no autograd flags)."""
    return {k: v.detach() for k, v in make_params(grad=False).items()}


# parameters whose gradient is identically zero in exact arithmetic (bias in front of a
# train-mode BatchNorm; key bias in front of a softmax): only rounding noise, never compared.
NULL_GRAD = (
    "net.map_encoder.cnn.0.bias", "net.map_encoder.cnn.3.bias", "net.map_encoder.cnn.6.bias",
    "net.map_decoder.layer0_1x1.0.bias", "net.map_decoder.layer1_1x1.0.bias", "net.map_decoder.conv_up0.0.bias",
    "net.map_decoder.conv_original_size0.0.bias", "net.map_decoder.conv_original_size1.0.bias",
    "net.map_decoder.conv_original_size2.0.bias", "net.state_text_k_layer.bias", "net.text_map_k_layer.bias",
)
