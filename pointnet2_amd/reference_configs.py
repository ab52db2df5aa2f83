"""Operator instances of the five BASELINE.json configurations ("Per-config op
shapes"), derived from the reference's model files. One table for the parity tests
(tests/test_configs_gpu.py), the timing scripts (scripts/config_shapes.py) and bench.py's legs.

SA row: (label, b, n, npoint, [(radius, nsample), ...], c)   c = feature channels grouped besides xyz
FP row: (label, b, n, m, c)                                  three_nn(n unknown, m known), interpolate c
"""

SA_LEVELS = [
    # config 1: BASELINE "Single SA layer on CPU"
    ("cfg1 SA", 2, 1024, 256, [(0.2, 32)], 0),
    # config 2: models/pointnet2_cls_ssg.py:32-33
    ("cfg2 cls_ssg L1", 32, 1024, 512, [(0.2, 32)], 0),
    ("cfg2 cls_ssg L2", 32, 512, 128, [(0.4, 64)], 128),
    # config 3: models/pointnet2_cls_msg.py:27-28 (xyz + normals sliced like pointnet2_part_seg.py:22-23)
    ("cfg3 cls_msg L1", 32, 4096, 512, [(0.1, 16), (0.2, 32), (0.4, 128)], 3),
    ("cfg3 cls_msg L2", 32, 512, 128, [(0.2, 32), (0.4, 64), (0.8, 128)], 320),
    # config 4: models/pointnet2_part_seg.py:26-27
    ("cfg4 part_seg SA1", 16, 2048, 512, [(0.2, 64)], 3),
    ("cfg4 part_seg SA2", 16, 512, 128, [(0.4, 64)], 128),
    # config 5: models/pointnet2_sem_seg.py:28-31 (per-GPU batch 8)
    ("cfg5 sem_seg SA1", 8, 8192, 1024, [(0.1, 32)], 0),
    ("cfg5 sem_seg SA2", 8, 1024, 256, [(0.2, 32)], 64),
    ("cfg5 sem_seg SA3", 8, 256, 64, [(0.4, 32)], 128),
    ("cfg5 sem_seg SA4", 8, 64, 16, [(0.8, 32)], 256),
    # the metric shape (BASELINE.json "metric")
    ("metric SA", 32, 4096, 1024, [(0.2, 32)], 0),
]

FP_LEVELS = [
    # config 4: models/pointnet2_part_seg.py:31-33 (FP1 interpolates from the ONE point of group_all)
    ("cfg4 part_seg FP1", 16, 128, 1, 1024),
    ("cfg4 part_seg FP2", 16, 512, 128, 256),
    ("cfg4 part_seg FP3", 16, 2048, 512, 128),
    # config 5: models/pointnet2_sem_seg.py:34-37
    ("cfg5 sem_seg FP1", 8, 64, 16, 512),
    ("cfg5 sem_seg FP2", 8, 256, 64, 256),
    ("cfg5 sem_seg FP3", 8, 1024, 256, 256),
    ("cfg5 sem_seg FP4", 8, 8192, 1024, 128),
]

# gradient all-reduce buckets of the data-parallel configs (fp32 elements; SURVEY.md 8e: ~3.9 MB sem_seg,
# ~5.9 MB cls_ssg -- train_multi_gpu.py:91-126 averages every trainable variable)
GRAD_BUCKET_FLOATS = {"sem_seg": 970_000, "cls_ssg": 1_470_000}
