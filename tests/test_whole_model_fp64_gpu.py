"""Whole-model gradient agreement: the four reference networks evaluated once in float64 on the
GPU's own geometry (scripts/whole_model_fp64.py), and BOTH fp32 training paths -- the fused nodes (csrc/train_mlp.hip) and
the layer-by-layer torch path -- compared with it. The fused path's L2 error of the flat gradient bucket must not exceed
twice the layer-by-layer path's (both are fp32 evaluations of a graph with ~20 stacked batch norms; the float64 run says
which of them is responsible for how much of their mutual difference). Reference: utils/pointnet_util.py:87-229,
models/pointnet2_{cls_ssg,cls_msg,part_seg,sem_seg}.py."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("which", ["cls_ssg", "cls_msg", "part_seg", "sem_seg"])
def test_fused_training_gradients_are_as_close_to_float64_as_torchs(cuda, which):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import whole_model_fp64 as W
    from train_step_bench import MODELS
    spec = [m for m in MODELS if which in m[0]][0]
    row = W.run_model(*spec)
    f, u = row["fused"], row["layer_by_layer"]
    print(row)
    assert f["loss_rel_err"] <= 1e-5 and f["logits_rel_err"] <= 1e-3
    assert f["grad_l2_err"] <= 2.0 * u["grad_l2_err"] + 1e-6, row
