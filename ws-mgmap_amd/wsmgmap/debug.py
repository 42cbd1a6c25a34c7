"""Every A/B and diagnostic switch of the package in ONE place, read ONCE.

The product path has no per-call environment look-ups: `sw` is filled from the process environment when the package is imported,
and the code reads plain attributes of it.  Tests and tools flip a switch with `monkeypatch.setattr(debug.sw, name, value)` (or
set the variable before the import); `sw.reload()` re-reads the environment.  Defaults are what every reported number uses.
Switches of experiments that were measured and dropped (round 3: WSMG_WGRAD_STREAM, WSMG_WGRAD_REDUCE_STREAM, WSMG_CONV_CAT,
WSMG_RNN_EXCL, WSMG_WGRAD_DBG_SKIP, WSMG_LSTM_DW_CONTIG, ...) are gone together with their code; DESIGN.md section 7 keeps the
measurements.

    name                 environment variable        default  meaning
"""
import os

_TABLE = [
    # (attribute, variable, default, kind, meaning)
    ("recurrent_chunks", "WSMG_RECURRENT_CHUNKS", 4, int, "time chunks of the pipelined recurrent core (wsmgmap/recurrent.py); 0 = the staged route"),
    ("rows_gemm", "WSMG_ROWS_GEMM", True, bool, "the recurrent core's dense layers as one launch each (csrc/wsmg_rows_gemm.hip); 0: the GEMM library"),
    ("fp8_fused", "WSMG_FP8_FUSED", True, bool, "ops.attention_fp8_shared as one launch (wsmg_attn_fp8_mfma_fused) where its grid barrier is safe"),
    ("fp8_row_fused", "WSMG_FP8_ROW_FUSED", True, bool, "single-query fp8 attention (one token set per row) as one launch: fold + attention per row (round 6)"),
    ("recurrent_chain", "WSMG_RECURRENT_CHAIN", True, bool, "each recurrence of the pipelined core as ONE launch chained to the attention stage by device-side counters"),
    ("decoder_streams", "WSMG_DECODER_STREAMS", 1, int, "0: map decoder on one stream; 1: side stream unless ranks share a GPU; 2: always"),
    ("early_dedup", "WSMG_EARLY_DEDUP", True, bool, "instruction dedup on its own stream when the producer marked the tokens ready (ops.mark_inputs_ready)"),
    ("early_dedup_dp", "WSMG_EARLY_DEDUP_DP", False, bool, "the early dedup also under a process group (needs the exchange on a policy stream: GradAllReducer(exchange_stream=...))"),
    ("rollout_fold", "WSMG_ROLLOUT_FOLD", True, bool, "rollout map stack with BatchNorm-folded cached operands"),
    ("depth_engine", "WSMG_DEPTH_ENGINE", False, bool, "frozen depth ResNet50 on the bf16 NHWC engine (5 % error: opt-in)"),
    ("fused_cls_tail", "WSMG_FUSED_CLS_TAIL", True, bool, "classifier tail (BN + ReLU + 1x1 + CE + pool) in one pass per direction"),
    ("bn_fused_stats", "WSMG_BN_FUSED_STATS", True, bool, "BatchNorm sums in the producing convolution's epilogue"),
    ("relu_producer_mask", "WSMG_RELU_PRODUCER_MASK", True, bool, "fused-ReLU masks of map_encoded / map_classified_linear in map_cated_linear's backward-data epilogue (round 6)"),
    ("conv_into_cat", "WSMG_CONV_INTO_CAT", True, bool, "map_encoded / map_classified_linear write straight into their slices of the concatenation (round 6)"),
    ("strided_grads", "WSMG_STRIDED_GRADS", True, bool, "channel slices of a concatenation's gradient read in place"),
    ("wgrad_atomics", "WSMG_WGRAD_ATOMICS", False, bool, "weight gradients through float atomics instead of slabs + ordered reduce (not bit-reproducible)"),
    ("conv_splitk", "WSMG_CONV_SPLITK", True, bool, "split-K for rollout-size layers"),
    ("rows_linear", "WSMG_ROWS_LINEAR", True, bool, "one-launch dense layers for <= 16 rows"),
    ("bev_fused", "WSMG_BEV_FUSED", True, bool, "scatter + rotation in one launch, plane-consuming fuse"),
    ("bev_compact", "WSMG_BEV_COMPACT", True, bool, "BEV: the index launch packs the valid sources and the scatter walks only those (round 6)"),
    ("rnn_stock", "WSMG_RNN_STOCK", False, bool, "the three recurrences on the stock (MIOpen) GRU / LSTM: no persistent kernel at all (bench.py's last fallback; needs recurrent_chunks = 0)"),
    ("rnn_poison", "WSMG_RNN_POISON", False, bool, "NaN-fill the persistent kernels' workspaces first (stress tool)"),
    ("feeder_pin", "WSMG_FEEDER_PIN", True, bool, "register the feeder's shared-memory ring as pinned memory"),
    ("keep_blas", "WSMG_KEEP_BLAS", False, bool, "leave torch's BLAS backend choice alone (default: rocBLAS)"),
]


class Switches:
    def __init__(self):
        self.reload()

    def reload(self):
        for attr, var, default, kind, _ in _TABLE:
            raw = os.environ.get(var)
            if raw is None:
                val = default
            elif kind is bool:
                val = raw not in ("0", "", "false", "False")
            else:
                val = kind(raw)
            setattr(self, attr, val)
        return self

    def describe(self):
        return "\n".join("    %-20s %-27s %-8s %s" % (a, v, d, m) for a, v, d, _, m in _TABLE)


sw = Switches()
__doc__ += sw.describe() + "\n"
