"""MGMapNet: the multi-granularity-map policy network, host side.

Attribute tree, state_dict keys and the in-place contract (writes `observations['rgb_ego_map']`,
`rnn_hidden_states[...]`, `self.att_map_t_m`) follow the reference
(vlnce_baselines/models/mg_map_policy.py:19-251).  The data path is re-designed for MI355X:

  * the whole map stack (encoder, UNet decoder, classifier, the three map projections) runs
    NHWC in the gfx950 conv / batch-norm / pooling kernels (operator 2);
  * both cross-modal attentions run in the single-query attention kernel (operator 3) on
    token-major keys/values; the k=1 Conv1d key projections are MFMA GEMMs of the conv engine;
  * the BEV projection / scatter / global-map fuse run in the BEV kernels (operator 1);
  * instruction encoding is de-duplicated over the time axis of a teacher-forcing batch;
  * the two GRU state encoders run as persistent whole-sequence kernels (csrc/wsmg_rnn.hip);
  * the instruction LSTM cell, Linear heads and losses stay stock PyTorch-ROCm.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import debug, ops, recurrent
from ..common.rgb_mapping import RGBMapping
from .encoders.instruction_encoder import InstructionEncoder
from .encoders.map_encoder import MapDecoder, MapEncoder
from .encoders.resnet_encoders import VlnResnetDepthEncoder
from .encoders.unet_encoder import UNet
from .rnn_state_encoder import RNNStateEncoder

SEM_CLASSES = 27
SEM_PAD = 32  # channel padding of the 27-class tensors inside the NHWC engine


class MGMapNet(nn.Module):
    def __init__(self, observation_space, model_config):
        super().__init__()
        mc = model_config
        self.model_config = mc
        hid = mc.STATE_ENCODER.hidden_size
        self._hidden_size = hid
        self._inputs = list(mc.STATE_ENCODER.input_type)

        self.instruction_encoder = InstructionEncoder(mc.INSTRUCTION_ENCODER)

        self.rgb_encoder = UNet(mc)
        for p in self.rgb_encoder.parameters():
            p.requires_grad = False
        self.rgb_linear = nn.Sequential(nn.AdaptiveAvgPool1d(1), nn.Flatten(),
                                        nn.Linear(self.rgb_encoder.output_shape[0], mc.RGB_ENCODER.output_size), nn.ReLU(True))

        self.depth_encoder = VlnResnetDepthEncoder(observation_space, output_size=mc.DEPTH_ENCODER.output_size,
                                                   checkpoint=mc.DEPTH_ENCODER.ddppo_checkpoint,
                                                   backbone=mc.DEPTH_ENCODER.backbone, spatial_output=True)
        self.depth_linear = nn.Sequential(nn.Flatten(), nn.Linear(int(np.prod(self.depth_encoder.output_shape)),
                                                                  mc.DEPTH_ENCODER.output_size), nn.ReLU(True))

        self.rgb_mapping_module = RGBMapping(mc.RGBMAPPING)
        map_channel = mc.RGBMAPPING.map_depth
        self.map_encoder = MapEncoder(mc.MAP_ENCODER.ego_map_size, map_channel, mc.MAP_ENCODER.output_size)
        self.map_decoder = MapDecoder(mc.MAP_ENCODER.output_size)
        self.map_classfier = nn.Sequential(  # (sic) name is part of the checkpoint contract
            nn.ConvTranspose2d(self.map_decoder.output_shape[0], 32, kernel_size=4, stride=2, padding=1, bias=False),
            nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 32, kernel_size=3, stride=1, padding=1, bias=False),
            nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, SEM_CLASSES, kernel_size=1, stride=1, padding=0, bias=True),
        )
        msz = mc.MAP_ENCODER.output_size
        self.map_encoded_linear = nn.Sequential(nn.Conv2d(self.map_encoder.output_shape[0], 128, 3, stride=1, padding=1), nn.ReLU())
        self.map_classified_linear = nn.Sequential(nn.Conv2d(SEM_CLASSES, 128, 3, stride=1, padding=1), nn.ReLU())
        self.map_cated_linear = nn.Sequential(nn.Conv2d(128 * 2, msz, 3, stride=1, padding=1), nn.ReLU())
        self.map_linear = nn.Sequential(nn.AdaptiveAvgPool1d(1), nn.Flatten(), nn.Linear(msz, msz), nn.ReLU(True))

        first_in = ((mc.RGB_ENCODER.output_size if "rgb" in self._inputs else 0)
                    + (mc.DEPTH_ENCODER.output_size if "depth" in self._inputs else 0)
                    + (msz if "map" in self._inputs else 0))
        self.state_encoder = RNNStateEncoder(first_in, hid, 1, mc.STATE_ENCODER.rnn_type)

        self.state_text_q_layer = nn.Linear(hid, hid // 2)
        self.state_text_k_layer = nn.Conv1d(self.instruction_encoder.output_size, hid // 2, 1)
        self.text_map_q_layer = nn.Linear(self.instruction_encoder.output_size, hid // 2)
        self.text_map_k_layer = nn.Conv1d(self.map_encoder.output_shape[0], hid // 2, 1)
        self.register_buffer("_scale", torch.tensor(1.0 / ((hid // 2) ** 0.5)))
        self._scale_f = 1.0 / ((hid // 2) ** 0.5)

        second_in = hid + hid // 2 + (hid // 2 if "map" in self._inputs else 0)
        self.second_state_compress = nn.Sequential(nn.Linear(second_in, hid), nn.ReLU(True))
        self.second_state_encoder = RNNStateEncoder(hid, hid, 1, mc.STATE_ENCODER.rnn_type)
        self._output_size = hid
        self.att_map_t_m = None
        self._side_stream = None
        # time chunks of the pipelined recurrent core of the update path (wsmgmap/recurrent.py); 0 = the staged route
        self.recurrent_chunks = int(debug.sw.recurrent_chunks)
        self.skip_pred_map_nchw = False   # set by BasePolicy around its own forward: it consumes sem_logits_nhwc
        self.sem_logits_nhwc = None
        self.sem_ce_rows = None           # the prediction monitor's per-sample loss when the fused classifier tail computed it
        self._gt_semantic_map = None      # set by BasePolicy.forward for that loss (observations['gt_semantic_map'])
        self._cls_tail_allowed = False    # likewise: the fused tail's logits carry no gradient
        # storage type of the map-stack activations: float32 (parity mode, f32 MFMA) or bfloat16
        # (BASELINE configs[1]; bf16 MFMA, float32 accumulation, float32 master weights)
        # (not a reference field; both spellings are accepted: `COMPUTE_DTYPE` as yacs nodes are usually written and
        # `compute_dtype` as default_model_config's keyword is called; anything but f32 / bf16 is an error, not a silent f32)
        want = getattr(mc, "COMPUTE_DTYPE", None)
        if want is None:
            want = getattr(mc, "compute_dtype", "f32")
        want = str(want).lower()
        # "bf16+f32grad" (round 6, opt-in): bf16 as above, but the weight gradients of the FIRST layer of each backward chain —
        # map_encoder.cnn.0, map_decoder.base_model.conv1, map_decoder.conv_original_size0: the tensors that drift most from a
        # float32 run — are taken from a 16-mantissa-bit dY (hi + lo bf16 pair): two weight-gradient launches each
        self.f32grad = want in ("bf16+f32grad", "bfloat16+f32grad")
        if self.f32grad:
            want = "bf16"
        if want not in ("f32", "fp32", "float32", "bf16", "bfloat16"):
            raise ValueError(f"MODEL.COMPUTE_DTYPE must be 'f32', 'bf16' or 'bf16+f32grad', got {want!r}")
        self.compute_dtype = torch.bfloat16 if want in ("bf16", "bfloat16") else torch.float32
        if self.f32grad:
            for conv in (self.map_encoder.cnn[0], self.map_decoder.base_model.conv1, self.map_decoder.conv_original_size0[0]):
                conv._wsmg_f32grad = True
        if self.compute_dtype == torch.bfloat16:   # the frozen RGB UNet follows: bf16 NHWC engine on the rollout path
            self.rgb_encoder.base_model.engine_dtype = torch.bfloat16
            # the depth ResNet50 stays float32 by default (ddppo_resnet.py: bf16 storage compounds over its 53 GroupNorm layers)
            if debug.sw.depth_engine and hasattr(self.depth_encoder.visual_encoder, "engine_dtype"):
                self.depth_encoder.visual_encoder.engine_dtype = torch.bfloat16

        self.train()
        self.depth_encoder.eval()
        self.rgb_encoder.eval()

    # -- habitat `Net` properties ------------------------------------------------
    @property
    def output_size(self):
        return self._output_size

    @property
    def is_blind(self):
        return False

    @property
    def num_recurrent_layers(self):
        return self.state_encoder.num_recurrent_layers + self.second_state_encoder.num_recurrent_layers

    # -- operator 3 --------------------------------------------------------------
    def _key_projection(self, conv1d: nn.Conv1d, tokens_tm):
        """Conv1d(k=1) over a token-major [B, I, C] tensor = 1x1 conv of the NHWC engine."""
        b, i, c = tokens_tm.shape
        w = conv1d.weight.unsqueeze(-1)  # [O, C, 1, 1]
        return ops.conv2d(tokens_tm.view(b, 1, i, c), w, conv1d.bias, 1, 0).view(b, i, -1)

    def _attn(self, q, k, v, mask=None):
        """q [B,C]; k, v [B,I,C] token-major -> (context [B,C], weights [B,I])."""
        return ops.attention(q.contiguous(), k.contiguous(), v.contiguous(), mask, self._scale_f)

    # -- operator 2 --------------------------------------------------------------
    def _ego_to_nhwc(self, ego_map):
        c = ego_map.shape[1]
        cpad = (c + 31) // 32 * 32   # the conv engine works on multiples of 32 channels (cfg4: 40 -> 64)
        nhwc_view = ego_map.permute(0, 2, 3, 1)
        if nhwc_view.is_contiguous() and cpad == c:
            # channels-last storage: what our BEV kernels emit (float32) and what DeviceCollator(ego_map_nhwc_bf16=True) emits
            # (already bf16: nothing to do — the feeder route has no layout / dtype pass at all)
            return nhwc_view if nhwc_view.dtype == self.compute_dtype else nhwc_view.to(self.compute_dtype)
        return ops.to_nhwc(ego_map.float().contiguous(), cpad, dtype=self.compute_dtype)

    _token_sink = None

    def map_stack(self, ego_map):
        """ego map [B,C,E,E] -> (map tokens [B, S*S, 256] token-major, pred_sem_map [B,27,2S,2S])."""
        from .encoders.map_encoder import batched_bumps
        with batched_bumps():
            return self._map_stack(ego_map)

    def _map_stack_conv_weights(self):
        d, c = self.map_decoder, self.map_classfier
        stem = d.base_model   # (its layer2-4 / fc exist for the checkpoint contract only and are never run)
        ws = [self.map_encoder.cnn[i].weight for i in (0, 3, 6)]
        ws += [stem.conv1.weight] + [w for blk in stem.layer1 for w in (blk.conv1.weight, blk.conv2.weight)]
        ws += [seq[0].weight for seq in (d.layer0_1x1, d.layer1_1x1, d.conv_up0, d.conv_original_size0, d.conv_original_size1,
                                         d.conv_original_size2)]
        return ws + [c[0].weight, c[3].weight, self.map_encoded_linear[0].weight, self.map_classified_linear[0].weight,
                     self.map_cated_linear[0].weight]

    _fold = None   # FoldCache of the rollout route, created on first use

    def refresh_folded(self):
        """Re-fold (in place) the rollout route's cached convolution operands whose parameters changed; GraphedAct calls
        this before every replay.  Returns the number of layers re-folded."""
        n = 0
        for cache in (self._fold, getattr(getattr(self.rgb_encoder, "base_model", None), "_fold_cache", None),
                      getattr(getattr(self.depth_encoder, "visual_encoder", None), "_wcache", None)):
            if cache is not None:
                n += cache.refresh()
        return n + self.instruction_encoder.packed_lstm_weights(refresh_only=True)

    def _map_stack_rollout(self, ego_map):
        """The map stack in eval mode without autograd (the rollout step), bf16: every convolution takes cached, BatchNorm-
        folded OHWI operands (encoders.map_encoder.FoldCache) — one launch per conv + BN + ReLU, no per-step weight
        re-layout.  Same arithmetic as _map_stack up to the bf16 rounding of the folded weights."""
        from .encoders.map_encoder import FoldCache, conv_infer, conv_transpose_infer
        if self._fold is None:
            self._fold = FoldCache()
        f = self._fold
        x = self._ego_to_nhwc(ego_map)
        enc = self.map_encoder(x, fold=f)
        self._encoder_done = torch.cuda.Event()
        self._encoder_done.record(torch.cuda.current_stream())
        enc_proj = conv_infer(enc, f, self.map_encoded_linear[0])
        dec = self.map_decoder(enc, fold=f)
        c = self.map_classfier
        y = conv_transpose_infer(dec, f, c[0], c[1])
        y = conv_infer(y, f, c[3], c[4])
        sem = conv_infer(y, f, c[6], relu=False, cout_pad=SEM_PAD)                   # [B,2S,2S,32], channels 27.. are 0
        self.sem_logits_nhwc = sem
        pred_sem_map = None if self.skip_pred_map_nchw else ops.to_nchw(sem, SEM_CLASSES)
        cls_proj = conv_infer(ops.avgpool2(sem), f, self.map_classified_linear[0])
        emb = conv_infer([enc_proj, cls_proj], f, self.map_cated_linear[0])
        b, s1, s2, ch = emb.shape
        return emb.view(b, s1 * s2, ch), pred_sem_map

    def _map_stack(self, ego_map):
        train = self.training
        if (not train and not torch.is_grad_enabled() and ego_map.is_cuda and self.compute_dtype == torch.bfloat16
                and debug.sw.rollout_fold):
            return self._map_stack_rollout(ego_map)
        laid = getattr(self, "_laid_event", None)             # forward() already queued the layout (update path)
        self._laid_event = None
        if ego_map.is_cuda and laid is None:
            # operands of all the map stack's convolutions in one launch — on the instruction branch's stream, beside the ego
            # map's NCHW -> NHWC conversion (0.37 ms of pure HBM traffic) instead of in front of it (0.06 ms)
            entry = getattr(self, "_entry_event", None)
            if entry is not None and self.recurrent_chunks > 0 and not torch.cuda.is_current_stream_capturing():
                if self._side_stream is None:
                    self._side_stream = ops.helper_stream("instruction")
                self._side_stream.wait_event(entry)          # the optimizer's writes to the parameters are complete there
                with torch.cuda.stream(self._side_stream):
                    ops.prelayout_conv_weights(self._map_stack_conv_weights(), self.compute_dtype)
                laid = torch.cuda.Event()
                laid.record(self._side_stream)
            else:
                ops.prelayout_conv_weights(self._map_stack_conv_weights(), self.compute_dtype)
        x = self._ego_to_nhwc(ego_map)
        if laid is not None:
            torch.cuda.current_stream().wait_event(laid)
        enc = self.map_encoder(x)
        # the stem's forward is a PERSISTENT kernel, one workgroup per CU: the instruction branch's persistent LSTM (16 workgroups
        # that claim their CUs) must not start beside it, or 16 of the stem's workgroups wait for the LSTM to finish and then
        # do their whole share alone (_encode_instruction waits for this event before the LSTM launch)
        self._encoder_done = None
        if ego_map.is_cuda:
            self._encoder_done = torch.cuda.Event()
            self._encoder_done.record(torch.cuda.current_stream())
        conv = lambda t, seq, pad: ops.conv2d(t, seq[0].weight, seq[0].bias, 1, pad, relu=True)  # noqa: E731
        # the encoded map has three consumers: their gradients meet in one launch (ops.fanout3) instead of two autograd adds
        e_tok, e_full, e_stem = ops.fanout3(enc) if (train and enc.dtype == torch.bfloat16) else (enc, enc, enc)
        # map_encoded_linear (0.17 ms forward, 0.38 ms backward of full-chip kernels) depends on the encoded map only and is consumed
        # after the decoder: it goes on the decoder's side stream, in front of the full-resolution branch (round 4).  Forward it
        # then runs beside the stem's convolution; backward — autograd replays a node on its forward stream — its two kernels run
        # beside the resnet branch's ~45 launch-latency-bound kernels instead of alone on the main stream after them.
        side = self.map_decoder.side_stream(enc) if (enc.is_cuda and not torch.cuda.is_current_stream_capturing()) else None
        # round 6: the two projections write straight into their channel slices of the tensor map_cated_linear reads (torch.cat at
        # mg_map_policy.py:197 of the reference: no concatenation pass), and map_cated_linear's input-gradient kernel applies their
        # fused ReLUs' masks and hands each its own contiguous part (ops.conv2d_over_written_parts)
        cat_base = rs_e = rs_c = None
        co_e, co_c = self.map_encoded_linear[0].out_channels, self.map_classified_linear[0].out_channels
        if (train and enc.dtype == torch.bfloat16 and debug.sw.conv_into_cat and torch.is_grad_enabled() and co_e % 8 == 0 and co_c % 8 == 0
                and self.map_cated_linear[0].kernel_size == (3, 3)):
            cat_base = torch.empty(enc.shape[0], enc.shape[1], enc.shape[2], co_e + co_c, device=enc.device, dtype=enc.dtype)
            rs_e, rs_c = ops.ReluSink(), ops.ReluSink()
        conv_e = lambda t: ops.conv2d(t, self.map_encoded_linear[0].weight, self.map_encoded_linear[0].bias, 1, 1, relu=True,  # noqa: E731
                                      relu_sink=rs_e, into=None if cat_base is None else (cat_base, 0))
        if side is not None:
            main_s = torch.cuda.current_stream()
            side.wait_stream(main_s)
            e_tok.record_stream(side)
            if cat_base is not None:
                cat_base.record_stream(side)
            with torch.cuda.stream(side):
                enc_proj = conv_e(e_tok)
        else:
            enc_proj = conv_e(e_tok)
        dec = self.map_decoder((e_full, e_stem))      # (joins the side stream into the main one before its last convolution)
        if side is not None:
            enc_proj.record_stream(main_s)
        c = self.map_classfier
        from .encoders.map_encoder import bump
        fused = train and dec.dtype == torch.bfloat16
        st1 = ops.bn_stats_slabs(id(c[1]), 32, dec.device) if fused else None
        y = ops.conv_transpose2d(dec, c[0].weight, 2, 1, st1)
        bump(c[1], train)
        y = ops.bn_act(y, c[1].weight, c[1].bias, c[1].running_mean, c[1].running_var, train, True, None, c[1].momentum, c[1].eps, st1)
        st4 = ops.bn_stats_slabs(id(c[4]), 32, dec.device) if fused else None
        y = ops.conv2d(y, c[3].weight, None, 1, 1, stats=st4)
        bump(c[4], train)
        self.sem_ce_rows = None
        # (only under BasePolicy.forward, which takes the loss from `sem_ce_rows` and never differentiates the logits themselves:
        #  `skip_pred_map_nchw` is its mark; a caller that builds its own loss on pred_sem_map gets the unfused, differentiable route)
        if (st4 is not None and self.skip_pred_map_nchw and getattr(self, "_cls_tail_allowed", False) and SEM_PAD == 32 and ops.cls_tail_ok(y, SEM_CLASSES) and c[6].bias is not None):
            # BatchNorm + ReLU + the 1 x 1 convolution + the prediction monitor's cross-entropy + the 2 x 2 average pool in ONE pass
            # per direction over the 48 x 48 x 32 activation (csrc/wsmg_cls_tail.hip) instead of four forward / six backward passes
            sem, pooled, self.sem_ce_rows = ops.cls_tail(y, st4, c[4], c[6], getattr(self, "_gt_semantic_map", None))
        else:
            y = ops.bn_act(y, c[4].weight, c[4].bias, c[4].running_mean, c[4].running_var, train, True, None, c[4].momentum, c[4].eps, st4)
            pad_o = SEM_PAD - SEM_CLASSES
            w6 = torch.nn.functional.pad(c[6].weight, (0, 0, 0, 0, 0, 0, 0, pad_o))   # [32,32,1,1]
            b6 = torch.nn.functional.pad(c[6].bias, (0, pad_o))
            sem = ops.conv2d(y, w6, b6, 1, 0)                                        # [B,2S,2S,32], channels 27.. are 0
            pooled = ops.avgpool2(sem)
        if fused:
            ops.bn_stats_done(id(c[1]), 32, dec.device)
            ops.bn_stats_done(id(c[4]), 32, dec.device)
        # the loss reads the NHWC logits directly (policy.aux_prediction); the reference-layout [B,27,2S,2S] tensor is
        # only materialised for callers that ask for it
        self.sem_logits_nhwc = sem
        pred_sem_map = None if self.skip_pred_map_nchw else ops.to_nchw(sem, SEM_CLASSES)
        # (27 -> 32 input channels: ops.conv2d zero-pads the weight to the activation's channel count)
        cls_proj = ops.conv2d(pooled, self.map_classified_linear[0].weight, self.map_classified_linear[0].bias, 1, 1, relu=True,
                              relu_sink=rs_c, into=None if cat_base is None else (cat_base, co_e))
        if cat_base is not None:
            emb = ops.conv2d_over_written_parts(enc_proj, cls_proj, cat_base, self.map_cated_linear[0].weight, self.map_cated_linear[0].bias,
                                                relu=True, relu_sink=self._token_sink, mask_sinks=[rs_e, rs_c])
        else:
            emb = ops.conv2d_cat([enc_proj, cls_proj], self.map_cated_linear[0].weight, self.map_cated_linear[0].bias, 1, 1, relu=True,
                                 relu_sink=self._token_sink)
        b, s1, s2, ch = emb.shape
        return emb.view(b, s1 * s2, ch), pred_sem_map

    # -- forward -------------------------------------------------------------------
    def _encode_instruction(self, observations, entry):
        """Instruction branch (dedup + packed bi-LSTM + key projection) on a side stream: it is
        independent of the map stack, its persistent 16-workgroup kernels leave 94 % of the CUs free,
        and autograd replays its backward on the same stream — so it overlaps the convolutions in
        both directions.  Joined (event wait) right before the text attention.

        `entry` is an event recorded on the main stream when forward() was entered: the side stream waits for
        that (its inputs — the tokens and the parameters — are complete there), not for the map stack, and this
        method is CALLED after the map stack has been queued: the dedup needs one host read-back (the number of
        unique instructions), and while the host waits for it the GPU still has the map-stack forward to run,
        instead of draining (measured: 0.9 ms of idle GPU per update when the read-back came first)."""
        if self._side_stream is None:
            self._side_stream = ops.helper_stream("instruction")
        side = self._side_stream
        side.wait_event(entry)
        tok = observations["instruction"]
        dd = observations.get("instruction_dedup")
        if dd is None and torch.is_tensor(tok):
            dd = ops.attached_instruction_dedup(tok)      # computed on the host by the feeder's collate: no kernel, no read-back
            if dd is not None:
                # (made on the collate stream: this branch's stream waits for the producer's event, and the persistent kernels'
                #  status word — otherwise read behind the dedup's read-back — is read here)
                ev = ops.inputs_ready_event(tok)
                if ev is not None:
                    side.wait_event(ev)
                for t in dd:
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(side)
                ops.check_rnn_status()
        # (not under a process group: with the host free to run ahead of the GPU, the data-parallel bench path — one-rank RCCL
        #  communicator, the only form a 1-GPU box can host — ran 17.7 ms per update instead of 11.3, on the GPU's own clock
        #  (profiles/r04_dp1_early_dedup.txt); not understood, so the exchange keeps the round-3 behaviour: one read-back per update
        #  that waits for the previous update)
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() and not debug.sw.early_dedup_dp
        ready = ops.inputs_ready_event(tok) if (dd is None and tok.is_cuda and torch.is_grad_enabled() and debug.sw.early_dedup
                                                and not multi and not torch.cuda.is_current_stream_capturing()) else None
        if ready is not None:
            # the producer of the tokens told us when they were complete (ops.mark_inputs_ready): the dedup — parameter-free — runs on a
            # stream that waits for that alone, and its read-back does not wait for the previous update (see ops/core.py)
            if getattr(self, "_early_stream", None) is None:
                self._early_stream = ops.helper_stream("early", priority=-1)
            early_s = self._early_stream
            early_s.wait_event(ready)
            with torch.cuda.stream(early_s):
                tok.record_stream(early_s)
                dd = self.instruction_encoder.dedup(tok)
            side.wait_stream(early_s)
            for t in dd:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(side)
        with torch.cuda.stream(side):
            after = getattr(self, "_encoder_done", None)
            instr_u, mask_u, inverse = self.instruction_encoder.encode_unique(tok, dedup=dd, lstm_after=after)
            text_k_u = self._key_projection(self.state_text_k_layer, instr_u)
            # the B rows attend over the U unique sets in place (ops.attention_shared): no per-row copies
            text = (text_k_u.contiguous(), instr_u.contiguous(), mask_u.to(torch.uint8).contiguous(), inverse.contiguous())
        return text, side

    def forward(self, observations, rnn_hidden_states, prev_actions, masks):
        ops.reset_pass_state()
        entry = torch.cuda.Event()
        entry.record(torch.cuda.current_stream())
        self._entry_event = entry
        ops.mark("entry")
        # Update path: the map stack's weight operands are laid out FIRST on the instruction branch's stream.  Queued from
        # _map_stack they sat behind the cached features' dense layers (queued below, on the same stream, and stretched to 0.37 ms by
        # the ego map's layout conversion they run beside): the stem's convolution started 0.15 ms after its input was ready.
        self._laid_event = None
        ego = observations.get("rgb_ego_map")
        if (torch.is_grad_enabled() and torch.is_tensor(ego) and ego.is_cuda and self.recurrent_chunks > 0
                and not torch.cuda.is_current_stream_capturing()):
            if self._side_stream is None:
                self._side_stream = ops.helper_stream("instruction")
            self._side_stream.wait_event(entry)              # the optimizer's writes to the parameters are complete there
            with torch.cuda.stream(self._side_stream):
                ops.prelayout_conv_weights(self._map_stack_conv_weights(), self.compute_dtype)
            self._laid_event = torch.cuda.Event()
            self._laid_event.record(self._side_stream)
        # Rollout (no autograd, RGB encoded from pixels): the instruction branch is queued FIRST and runs beside the frozen
        # RGB encoder — in a captured step (graph.GraphedAct) it otherwise lands behind the map decoder's side branch and the
        # main stream idles through the whole 0.45 ms LSTM (B = 1).  Training keeps the order described in _encode_instruction.
        early = not torch.is_grad_enabled() and "rgb_features" not in observations and observations["instruction"].is_cuda
        if early:
            self._encoder_done = None
            text, side = self._encode_instruction(observations, entry)
        rgb_embedding, rgb_embedding_proj = self.rgb_encoder(observations)
        depth_embedding = self.depth_encoder(observations)
        rows = ops.rows_route(rgb_embedding.float())   # rollout: every dense layer below is one launch (ops.linear_rows)
        lin = lambda m, x, act=None: ops.linear_rows(x, m.weight, m.bias, act) if rows else m(x)  # noqa: E731

        def dense_inputs():
            """The rgb / depth parts of the first state encoder's input (mg_map_policy.py:209-216 of the reference)."""
            out = []
            if "rgb" in self._inputs:
                if rows:   # AdaptiveAvgPool1d(1) + Flatten + Linear + ReLU
                    b, c = rgb_embedding.shape[:2]
                    out.append(ops.linear_rows(rgb_embedding.float().reshape(b, c, -1), self.rgb_linear[2].weight, self.rgb_linear[2].bias,
                                               "relu", pool=int(np.prod(rgb_embedding.shape[2:]))))
                else:
                    feat = torch.flatten(rgb_embedding.float(), 2)
                    if feat.is_cuda and not feat.requires_grad and feat.is_contiguous() and feat.shape[-1] <= 160:
                        # AdaptiveAvgPool1d(1) + Flatten as one coalesced pass (torch's reduction over a 49-long innermost axis reads
                        # the 51 MB feature at 1 TB/s: 50 us); the Linear + ReLU stay the module's
                        out.append(self.rgb_linear[3](self.rgb_linear[2](ops.mean_last(feat))))
                    else:
                        out.append(self.rgb_linear(feat))
            if "depth" in self._inputs:
                if rows:   # Flatten + Linear + ReLU
                    out.append(lin(self.depth_linear[1], torch.flatten(depth_embedding.float(), 1), "relu"))
                else:
                    out.append(self.depth_linear(torch.flatten(depth_embedding.float(), 2)))
            return out

        # Update path: these two depend on the cached features only, not on the map stack — they (and, under autograd, their
        # backward: leaves of the graph) run on the instruction branch's stream beside the map stack instead of on the critical
        # path between the map stack and the first recurrence (8 + 12 dependent launches of 5-20 us)
        # (not under a HIP-graph capture: wsmgmap.graph captures the one-stream form of these three — this, the weight layout
        #  and the pipelined recurrent core; the end of a capture that held the three-stream form crashed in the runtime)
        capturing = rgb_embedding.is_cuda and torch.cuda.is_current_stream_capturing()
        # (only with BOTH feature sets cached: from raw pixels the frozen encoders run on the main stream after `entry`)
        dense_early = (torch.is_grad_enabled() and not rows and not early and rgb_embedding.is_cuda and self.recurrent_chunks > 0
                       and not capturing and "rgb_features" in observations and "depth_features" in observations)
        if dense_early:
            if self._side_stream is None:
                self._side_stream = ops.helper_stream("instruction")
            # depth_embedding is WRITTEN on the main stream after `entry` (the spatial-embedding concatenation of the depth
            # encoder, also with cached features): the side stream waits for an event recorded behind it, not for `entry`
            # (read-before-write), and the caching allocator is told about the second stream
            encoders_done = torch.cuda.Event()
            encoders_done.record(torch.cuda.current_stream())
            self._side_stream.wait_event(encoders_done)
            for t in (rgb_embedding, depth_embedding):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self._side_stream)
            with torch.cuda.stream(self._side_stream):
                state_in = dense_inputs()
            dense_ready = torch.cuda.Event()
            dense_ready.record(self._side_stream)

        self.rgb_mapping_module(rgb_embedding_proj, observations, masks)
        # the map tokens feed their mean (state input) and the map attention: one merged, ReLU-masked gradient pass (TokenGradSink)
        self._token_sink = ops.TokenGradSink() if (torch.is_grad_enabled() and "map" in self._inputs) else None
        sink = self._token_sink
        map_tokens, pred_sem_map = self.map_stack(observations["rgb_ego_map"])
        self._token_sink = None
        self._entry_event = None      # (a direct map_stack() call outside forward() lays its weights out on its own stream)
        ops.mark("map_stack", map_tokens)
        if not early:
            text, side = self._encode_instruction(observations, entry)   # queued after the map stack, runs beside it

        if dense_early:
            cur = torch.cuda.current_stream()
            cur.wait_event(dense_ready)
            for t in state_in:
                t.record_stream(cur)
        else:
            state_in = dense_inputs()
        if "map" in self._inputs:
            tm = ops.token_mean(map_tokens, sink)
            state_in.append(lin(self.map_linear[2], tm, "relu") if rows else self.map_linear[3](self.map_linear[2](tm)))
        state_in = torch.cat(state_in, dim=1)

        n1 = self.state_encoder.num_recurrent_layers
        ops.mark("state_in", state_in)
        n_env = rnn_hidden_states.size(1)
        if (self.recurrent_chunks > 0 and not ops.ranks_share_gpu() and torch.is_grad_enabled() and not rows and "map" in self._inputs and n1 == 1
                and rnn_hidden_states.size(0) == 2 and recurrent.usable(state_in, map_tokens, n_env, text)):
            # GRU 1 -> text attention -> map attention -> compress -> GRU 2 as one autograd node, pipelined over time chunks on three
            # streams, parameter gradients off the chain (wsmgmap/recurrent.py); same kernels, same arithmetic row for row
            text_ready = torch.cuda.Event()
            text_ready.record(side)
            main = torch.cuda.current_stream()
            for t in text:
                t.record_stream(main)
            x, self.att_map_t_m, h1n, h2n = recurrent.recurrent_block(
                state_in, map_tokens, text, masks, rnn_hidden_states[0], rnn_hidden_states[1], self, n_env,
                chunks=self.recurrent_chunks, sink=sink, text_ready=text_ready,
                streams=None if capturing else (side, getattr(self.map_decoder, "_side", None)))
            rnn_hidden_states[0:n1] = h1n
            rnn_hidden_states[n1:] = h2n
            ops.mark("gru2", x)
            return x, rnn_hidden_states, pred_sem_map
        state, rnn_hidden_states[0:n1] = self.state_encoder(state_in, rnn_hidden_states[0:n1], masks)
        ops.mark("gru1", state)

        # instruction attention: keys projected once per unique instruction, gathered per row
        torch.cuda.current_stream().wait_stream(side)
        text_k_u, text_v_u, text_mask_u, inverse = text
        for t in text:
            t.record_stream(torch.cuda.current_stream())
        text_embedding, _ = ops.attention_shared(lin(self.state_text_q_layer, state).contiguous(), text_k_u, text_v_u, text_mask_u,
                                                 inverse, self._scale_f)

        # map attention
        # text_map_k_layer is folded into the query (ops._AttnFolded): the 576 map tokens are read once, as
        # keys and values, and no projected key tensor exists
        map_embedding, self.att_map_t_m = ops.attention_folded(
            lin(self.text_map_q_layer, text_embedding).contiguous(), self.text_map_k_layer.weight, self.text_map_k_layer.bias,
            map_tokens.contiguous(), None, self._scale_f, sink)

        parts = [state, text_embedding] + ([map_embedding] if "map" in self._inputs else [])
        x = torch.cat(parts, dim=1)
        x = lin(self.second_state_compress[0], x, "relu") if rows else self.second_state_compress(x)
        ops.mark("attention", x)
        x, rnn_hidden_states[n1:] = self.second_state_encoder(x, rnn_hidden_states[n1:], masks)
        ops.mark("gru2", x)
        return x, rnn_hidden_states, pred_sem_map
