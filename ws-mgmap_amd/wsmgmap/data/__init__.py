"""Trajectory-cache path in front of the policy update: record codec, sharded / length-sorted
iteration, and the device-side collate.  Mirrors vlnce_baselines/dagger_trainer.py:36-238."""
from .codec import (pack_record, unpack_record, change_data_type, pack_record_raw, recode_record, is_raw_record,  # noqa: F401
                    sparse_pack_ego, sparse_expand_ego, densify, has_sparse_ego)
from .dataset import TrajectoryDataset, block_shuffle, shard_range  # noqa: F401
from .collate import DeviceCollator, collate_fn  # noqa: F401
from .feeder import DeviceFeeder  # noqa: F401
