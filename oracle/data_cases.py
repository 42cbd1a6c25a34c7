"""TEST INFRASTRUCTURE (oracle) — seeded inputs of the trajectory-cache cases.

Shared by tools/make_goldens_data.py (which runs the UNMODIFIED reference on them), oracle/data_ref.py and
tests/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
All values come from np.random.RandomState with fixed seeds: identical on every machine."""
import numpy as np

# sensors of a cached trajectory with the on-disk dtypes of common_trainer.py:514-532 (small spatial sizes)
SENSORS = {
    "instruction": (np.int64, (6,)),
    "rgb_features": (np.float16, (4, 2, 2)),
    "depth_features": (np.float16, (3, 2, 2)),
    "rgb_ego_map": (np.float16, (4, 5, 5)),
    "gt_semantic_map": (np.int64, (5, 5)),
    "gt_path": (np.float16, (5, 5)),
    "progress": (np.float32, (1,)),
    "vln_oracle_action_sensor": (np.uint8, (1,)),
    "waypoint": (np.float32, (2,)),
}


def episode(seed, length):
    """(obs dict of [T, ...] arrays, prev_actions [T,2] f32, oracle_actions [T,2] f32)."""
    rng = np.random.RandomState(seed)
    obs = {}
    for name, (dt, shape) in SENSORS.items():
        if np.issubdtype(dt, np.integer):
            obs[name] = rng.randint(0, 27, size=(length,) + shape).astype(dt)
        else:
            obs[name] = (rng.randn(*((length,) + shape)) * 3).astype(dt)
    prev = rng.randn(length, 2).astype(np.float32)
    oracle = rng.randn(length, 2).astype(np.float32)
    return obs, prev, oracle


COLLATE_LENGTHS = [5, 2, 7]          # ragged batch of the collate case
LONG_LENGTHS = [203, 3]              # exercises the 200-step cap of collate_fn (dagger_trainer.py:82-83)
DATASET_LENGTHS = [3, 9, 4, 4, 7, 1, 12, 5, 5, 2, 8, 6, 3, 11, 4, 10, 2, 7, 9, 1, 6, 5, 8]   # 23 records
DATASET_CASES = [   # (world_size, rank, num_workers, worker_id, batch_size, python random seed)
    (1, 0, 0, 0, 4, 5),
    (2, 1, 0, 0, 4, 5),
    (2, 0, 2, 1, 3, 11),
]
