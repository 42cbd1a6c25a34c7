"""Model configuration: the `MODEL` sub-tree of the reference's yacs config
(vlnce_baselines/config/default.py:70-137) as a plain attribute dict, so the policy can be
constructed without habitat / yacs.  A habitat `Config` node works in its place unchanged
(only attribute access is used)."""


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def default_model_config(num_proc=1, gpu_id=0, ego_map_size=100, map_depth=64, global_map_size=240,
                         compute_dtype="f32"):
    return Cfg(
        COMPUTE_DTYPE=compute_dtype,  # "f32" (parity mode) or "bf16" — not a reference field
        INSTRUCTION_ENCODER=Cfg(vocab_size=2504, max_length=200, embedding_size=50, hidden_size=128, rnn_type="LSTM",
                                final_state_only=False, bidirectional=True, use_pretrained_embeddings=False,
                                embedding_file="", fine_tune_embeddings=False),
        RGB_ENCODER=Cfg(output_size=256, backbone="unet", pretrain_model=None),
        DEPTH_ENCODER=Cfg(output_size=128, backbone="resnet50", ddppo_checkpoint="NONE"),
        MAP_ENCODER=Cfg(ego_map_size=ego_map_size, output_size=256),
        STATE_ENCODER=Cfg(hidden_size=512, rnn_type="GRU", input_type=["rgb", "depth", "map"]),
        PROGRESS_MONITOR=Cfg(use=True, alpha=1.0),
        CONTRASTIVE_MONITOR=Cfg(use=True, alpha=1.0, target_tau=0.07),
        PREDICTION_MONITOR=Cfg(use=True, alpha=0.1),
        RGBMAPPING=Cfg(map_depth=map_depth, global_map_size=global_map_size, egocentric_map_size=ego_map_size,
                       resolution=0.12, gpu_id=gpu_id, num_proc=num_proc),
    )
