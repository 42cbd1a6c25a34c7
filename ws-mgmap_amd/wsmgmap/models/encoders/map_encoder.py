"""Map encoder (3 strided convs) and the UNet-style semantic-hallucination decoder over the
24 x 24 encoded map — operator 2 of the hot path.

Module / parameter names reproduce the reference's state_dict contract
(vlnce_baselines/models/encoders/map_encoder.py:16-112); the arithmetic runs in the gfx950
conv / batch-norm / pooling kernels of libwsmgmap.so on NHWC activations (wsmgmap.ops).
The nn.Conv2d / nn.BatchNorm2d children are parameter containers only.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from ... import debug, ops
from .resnet18 import ResNet18


_PENDING_BUMPS = None   # while a map-stack forward is collecting: the num_batches_tracked counters to advance


def bump(bn: nn.BatchNorm2d, train: bool):
    """nn.BatchNorm2d advances num_batches_tracked on every train-mode forward.  Inside `batched_bumps()` the counters
    are collected and advanced by ONE multi-tensor add at the end (16 one-element kernels per update otherwise)."""
    if train and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _PENDING_BUMPS is not None:
            _PENDING_BUMPS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1


class batched_bumps:
    def __enter__(self):
        global _PENDING_BUMPS
        self._outer = _PENDING_BUMPS
        _PENDING_BUMPS = []
        return self

    def __exit__(self, *exc):
        global _PENDING_BUMPS
        todo, _PENDING_BUMPS = _PENDING_BUMPS, self._outer
        if todo and exc[0] is None:
            with torch.no_grad():
                torch._foreach_add_(todo, 1)
        return False


def conv_bn_relu(x, conv: nn.Conv2d, bn: nn.BatchNorm2d, train: bool, relu=True, residual=None):
    """Conv2d -> BatchNorm2d (batch statistics when training) [-> + residual] -> ReLU, NHWC.
    A convolution marked `_wsmg_f32grad` (MGMapNet, COMPUTE_DTYPE = "bf16+f32grad": the first layer of each backward chain) takes its
    weight gradient from a 16-mantissa-bit dY (ops.GradLoSink)."""
    # (if x carries zero-padded channels, ops.conv2d pads the weight's input channels to match)
    if isinstance(x, (list, tuple)):   # convolution over a channel concatenation: one vectorised concatenation, then the conv
        # (running it part by part over the weight's input-channel slices was measured and dropped: DESIGN.md section 7)
        x = ops.cat_channels(x[0], x[1]) if len(x) == 2 else torch.cat(list(x), dim=-1)
    # bf16 training: the convolution's epilogue accumulates the BatchNorm sums of its output (no statistics pass over y)
    stats = None
    if train and x.dtype == torch.bfloat16 and conv.out_channels % 8 == 0:
        stats = ops.bn_stats_slabs(id(bn), conv.out_channels, x.device)
    lo = ops.GradLoSink() if (train and getattr(conv, "_wsmg_f32grad", False) and x.dtype == torch.bfloat16 and torch.is_grad_enabled()) else None
    y = ops.conv2d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], bias_grad_zero=train, stats=stats, lo_sink=lo)
    bump(bn, train)
    out = ops.bn_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, train, relu, residual, bn.momentum, bn.eps, stats, lo_sink=lo)
    if stats is not None:
        ops.bn_stats_done(id(bn), conv.out_channels, x.device)
    return out


class FoldCache:
    """Rollout (eval mode, no grad, bf16): the engine operands of a convolution — OHWI bf16 weight, float32 bias — with the
    eval-mode BatchNorm that follows it folded in, kept while the parameters and running statistics keep their versions.
    A stale entry is refreshed IN PLACE: a captured rollout graph (graph.GraphedAct) reads the same addresses at every
    replay, and `refresh()` before a replay is all it takes to follow an optimizer step.  Per rollout step this removes the
    weight re-layout launch and the two BatchNorm launches per layer of the unfolded route."""

    TRANSPOSED = -1     # `cin_pad` value that asks for the operands of an nn.ConvTranspose2d (conv_transpose_infer)

    def __init__(self):
        self.entries = {}

    @staticmethod
    def _versions(weight, bias, bn):
        v = (weight._version, -1 if bias is None else bias._version)
        if bn is not None:
            v += (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version)
        return v

    @staticmethod
    @torch.no_grad()
    def _fold(weight, bias, bn, cin_pad, cout_pad):
        if cin_pad == FoldCache.TRANSPOSED:     # nn.ConvTranspose2d parameter [Cin_t, Cout_t, KH, KW] -> IHWO = [Cout_t, KH, KW, Cin_t]
            w = weight.float()
            b = None if bias is None else bias.float()
            if bn is not None:
                scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                w = w * scale.view(1, -1, 1, 1)
                b = bn.bias.float() - bn.running_mean.float() * scale + (0 if b is None else b * scale)
            if b is not None and bias is not None and b.data_ptr() == bias.data_ptr():
                b = b.clone()
            return w.permute(1, 2, 3, 0).contiguous().to(torch.bfloat16), None if b is None else b.contiguous()
        w = weight.float()
        b = None if bias is None else bias.float()
        if bn is not None:
            scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
            w = w * scale.view(-1, 1, 1, 1)
            b = bn.bias.float() - bn.running_mean.float() * scale + (0 if b is None else b * scale)
        pad_i, pad_o = max(cin_pad - w.shape[1], 0), max(cout_pad - w.shape[0], 0)
        if pad_i or pad_o:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, pad_i, 0, pad_o))
            if b is not None and pad_o:
                b = torch.nn.functional.pad(b, (0, pad_o))
        if b is not None and bias is not None and b.data_ptr() == bias.data_ptr():
            b = b.clone()          # (a float32 bias without BatchNorm or padding would otherwise BE the parameter)
        return w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16), None if b is None else b.contiguous()

    def get(self, weight, bias, bn, cin_pad, cout_pad=0, owner=None):
        """owner: the module the parameters belong to — keys the entry, so a hit costs no parameter lookups on the module."""
        key = (id(weight) if owner is None else id(owner), cin_pad, cout_pad)
        e = self.entries.get(key)
        if e is None:
            if weight is None:
                weight, bias = owner.weight, owner.bias
            w, b = self._fold(weight, bias, bn, cin_pad, cout_pad)
            # (the entry keeps `owner` alive: its id, the key, cannot be handed to another module while the entry exists)
            e = self.entries[key] = [self._versions(weight, bias, bn), w, b, (weight, bias, bn, cin_pad, cout_pad), owner]
        else:
            src = e[3]
            ver = self._versions(src[0], src[1], src[2])
            if e[0] != ver:
                self._refresh(e, ver)
        return e[1], e[2]

    @torch.no_grad()
    def _refresh(self, e, ver):
        w, b = self._fold(*e[3])
        e[1].copy_(w)
        if b is not None:
            e[2].copy_(b)
        e[0] = ver

    def refresh(self):
        """Bring every entry up to date with its parameters (same storage).  Returns the number of entries re-folded."""
        n = 0
        for e in self.entries.values():
            ver = self._versions(*e[3][:3])
            if e[0] != ver:
                self._refresh(e, ver)
                n += 1
        return n


def conv_infer(x, cache: FoldCache, conv, bn=None, relu=True, residual=None, cout_pad=0):
    """One launch: conv (+ folded eval-mode BatchNorm) (+ ReLU) on a bf16 NHWC activation, operands from `cache`.
    With `residual`: relu(conv_bn(x) + residual) is written INTO `residual` by the convolution's epilogue (and returned)."""
    if isinstance(x, (list, tuple)):
        x = ops.cat_channels(x[0], x[1])
    w, b = cache.get(None, None, bn, x.shape[-1], cout_pad, owner=conv)
    return ops.conv2d_infer_bf16(x, w, b, conv.stride[0], conv.padding[0], relu, add_to=residual)


def conv_transpose_infer(x, cache: FoldCache, convt: nn.ConvTranspose2d, bn=None, relu=True):
    """One launch: ConvTranspose2d (+ folded eval-mode BatchNorm) (+ ReLU) on a bf16 NHWC activation, operands from `cache`."""
    w, b = cache.get(None, None, bn, FoldCache.TRANSPOSED, 0, owner=convt)
    return ops.conv_transpose2d_infer_bf16(x, w, b, convt.stride[0], convt.padding[0], relu)


def convrelu(in_channels, out_channels, kernel, padding):
    return nn.Sequential(
        nn.Conv2d(in_channels, out_channels, kernel, padding=padding),
        nn.BatchNorm2d(num_features=out_channels),
        nn.ReLU(inplace=True),
    )


def _out_dim(size, kernel, stride, padding):
    return int(np.floor((size + 2 * padding - (kernel - 1) - 1) / stride + 1))


class MapEncoder(nn.Module):
    _layers = [(8, 2, 3), (5, 2, 1), (3, 1, 1)]  # (kernel, stride, padding)

    def __init__(self, map_size, input_channel, output_channel):
        super().__init__()
        chans = [input_channel, 64, 128, output_channel]
        mods = []
        for (k, s, p), cin, cout in zip(self._layers, chans[:-1], chans[1:]):
            mods += [nn.Conv2d(cin, cout, k, stride=s, padding=p), nn.BatchNorm2d(num_features=cout), nn.ReLU(inplace=True)]
        self.cnn = nn.Sequential(*mods)
        d = map_size
        for k, s, p in self._layers:
            d = _out_dim(d, k, s, p)
        self.output_shape = [output_channel, d, d]

    def forward(self, x_nhwc, fold=None):
        """fold: a FoldCache => the rollout route (eval-mode BatchNorm folded into the convolutions, one launch a layer)."""
        train = self.training
        for i in (0, 3, 6):
            if fold is not None:
                x_nhwc = conv_infer(x_nhwc, fold, self.cnn[i], self.cnn[i + 1])
            else:
                x_nhwc = conv_bn_relu(x_nhwc, self.cnn[i], self.cnn[i + 1], train)
        return x_nhwc


class MapDecoder(nn.Module):
    """2-level UNet on a resnet18 stem (conv1(256->64,k7,s2), bn1, maxpool, layer1)."""

    def __init__(self, n_channel_in):
        super().__init__()
        self.base_model = ResNet18()
        self.base_model.conv1 = nn.Conv2d(n_channel_in, 64, kernel_size=7, stride=2, padding=3, bias=False)
        children = list(self.base_model.children())
        # same duplicate registrations as the reference => same state_dict keys
        self.layer0 = nn.Sequential(*children[:3])
        self.layer0_1x1 = convrelu(64, 64, 1, 0)
        self.layer1 = nn.Sequential(*children[3:5])
        self.layer1_1x1 = convrelu(64, 64, 1, 0)
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv_up0 = convrelu(64 + 64, 128, 3, 1)
        self.conv_original_size0 = convrelu(n_channel_in, 64, 3, 1)
        self.conv_original_size1 = convrelu(64, 64, 3, 1)
        self.conv_original_size2 = convrelu(64 + 128, 64, 3, 1)
        self.output_shape = [64, 100, 100]
        self._side = None

    def _block(self, x, blk, train, fold=None):
        if fold is not None:
            return conv_infer(conv_infer(x, fold, blk.conv1, blk.bn1), fold, blk.conv2, blk.bn2, residual=x)
        y = conv_bn_relu(x, blk.conv1, blk.bn1, train)
        return conv_bn_relu(y, blk.conv2, blk.bn2, train, relu=True, residual=x)

    def side_stream(self, x):
        """The stream the full-resolution branch runs on beside the resnet branch, or None for one stream (debug.sw.decoder_streams
        = 0, CPU tensors, ranks sharing a GPU)."""
        import torch
        # Under a process group the side stream is used when every rank of this node has a GPU to itself (ops.ranks_share_gpu: the
        # launcher's LOCAL_WORLD_SIZE <= visible GPUs, the only configuration of the target — one process per GPU, README.md:80-84
        # of the reference).  The stream is worth 0.25 ms per update (12.36 vs 12.62 ms, single process, one run).
        shared = ops.ranks_share_gpu()
        mode = int(debug.sw.decoder_streams)     # 0: one stream; 2: side stream even when ranks share a GPU (experiments)
        if not (x.is_cuda and mode != 0 and (not shared or mode == 2)):
            return None
        if self._side is None:
            self._side = ops.helper_stream("decoder")
        return self._side

    def forward(self, x, fold=None):
        import torch
        # x: the encoded map, or (alias for the full-resolution branch, alias for the stem) of it — ops.fanout3 in MGMapNet._map_stack
        x_full, x = x if isinstance(x, (tuple, list)) else (x, x)
        train = self.training
        if fold is not None:
            cr = lambda t, seq: conv_infer(t, fold, seq[0], seq[1])  # noqa: E731
        else:
            cr = lambda t, seq: conv_bn_relu(t, seq[0], seq[1], train)  # noqa: E731
        # The full-resolution branch (two 3x3 convs) is independent of the resnet branch until the last concatenation, and
        # the resnet branch is mostly launch-latency-bound (6x6 and 12x12 maps: ~15 us kernels that leave the chip idle):
        # the full-resolution branch runs on a side stream beside it — in backward too, where autograd replays every node
        # on its forward stream.  WSMG_DECODER_STREAMS=0: one stream.
        side = self.side_stream(x)
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            x_full.record_stream(side)   # saved for the side-stream backward of these layers: no reuse of its memory before that ran
            with torch.cuda.stream(side):
                x_original = cr(cr(x_full, self.conv_original_size0), self.conv_original_size1)
        else:
            x_original = cr(cr(x_full, self.conv_original_size0), self.conv_original_size1)
        stem = self.base_model
        layer0 = cr(x, (stem.conv1, stem.bn1))
        layer1 = ops.maxpool3x3s2(layer0)
        for blk in stem.layer1:
            layer1 = self._block(layer1, blk, train, fold)
        if fold is not None:   # rollout: upsample + concatenation in one launch
            up = cr(ops.upsample2x_cat(cr(layer1, self.layer1_1x1), cr(layer0, self.layer0_1x1)), self.conv_up0)
            if side is not None:
                main.wait_stream(side)
                x_original.record_stream(main)
            return cr(ops.upsample2x_cat(up, x_original), self.conv_original_size2)
        if x.dtype == torch.bfloat16:
            # upsample + torch.cat(dim=1) of the reference in one launch, under autograd too (ops._Up2Cat)
            up = cr(ops.upsample2x_cat(cr(layer1, self.layer1_1x1), cr(layer0, self.layer0_1x1)), self.conv_up0)
            if side is not None:
                main.wait_stream(side)
                x_original.record_stream(main)
            return cr(ops.upsample2x_cat(up, x_original), self.conv_original_size2)
        up = ops.upsample2x(cr(layer1, self.layer1_1x1))
        up = cr([up, cr(layer0, self.layer0_1x1)], self.conv_up0)        # torch.cat(dim=1) of the reference, folded into the conv
        up = ops.upsample2x(up)
        if side is not None:
            main.wait_stream(side)
            x_original.record_stream(main)
        return cr([up, x_original], self.conv_original_size2)
