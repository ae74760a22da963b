"""Host logic of the launch planner (no GPU work): which kernel `orv_gemm_bf16` would launch for the shapes of the denoise path at one, two and four
clips, and which operands the transformer keeps in the packed P16 layout.  The expectations are the measured choices (profiles/r5_gemm_t8_r192.txt,
r5_packed_qkv_b1.txt, r5_model_ab_packed*.txt); a change of the cost model that moves one of them should be a decision, not an accident."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orv_amd import ops  # noqa: E402
from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj  # noqa: E402

D, S = 1920, 3226


def test_headline_batch_keeps_the_256_row_tiles():
    M = 4 * S
    assert ops.gemm_kernel_name(M, 4 * D, D, 1, c_packed=True) == "gemm_t8_kernel<256, 1>"
    assert ops.gemm_kernel_name(M, 3 * D, D, 4) == "gemm_t8_kernel<256, 4> + gemm_t8_kernel<192, 0>"
    assert ops.gemm_kernel_name(M, D, 4 * D, 2, a_packed=True) == "gemm_d8_kernel<192, 2>"
    assert ops.gemm_kernel_name(M, D, D, 2, a_packed=True) == "gemm_d8_kernel<192, 2>"


def test_one_and_two_clips_take_the_192_row_tiles_where_they_fill_the_rounds():
    for B in (1, 2):
        M = B * S
        assert ops.gemm_kernel_name(M, 4 * D, D, 1, c_packed=True) == "gemm_t8r192_kernel<256, 1>", B      # 510 / 1020 tiles instead of 390 / 780
        assert ops.gemm_kernel_name(M, 2 * D, D, 4) == "gemm_t8r192_kernel<256, 4>", B                      # 255 / 510 instead of 195 / 390
    # packed C has ceil(M / 256) * 256 row slots: a 192-row tiling that would overrun them is not offered
    assert ops.gemm_kernel_name(1024, 4 * D, D, 1, c_packed=True) == "gemm_t8_kernel<256, 1>"


def test_packed_plan_per_batch():
    plan = CogVideoXTransformer3DModelTraj._packed_plan
    os.environ.pop("ORV_PACKED_QKV", None), os.environ.pop("ORV_GEMM_PACKED", None)
    p1, p2, p4 = plan(S, D), plan(2 * S, D), plan(4 * S, D)
    assert p1 == {"ffn": True, "out": True, "qkv": True, "ffn1": False}       # one clip: q | k | v in ONE d8 launch from the packed LayerNorm output
    assert p2 == {"ffn": True, "out": False, "qkv": False, "ffn1": False}
    assert p4 == {"ffn": True, "out": True, "qkv": False, "ffn1": False}
    # round 6: the packed GEMMs of one and two clips take the 192-row d8 tiles (255 / 510 tiles of 192 x 128 = full rounds for N = 1920, 510 of
    # 192 x 192 for q | k | v at one clip); the headline batch keeps the 256-row tiles (test above)
    assert ops.gemm_kernel_name(S, 3 * D, D, 4, a_packed=True) == "gemm_d8r192_kernel<192, 4>"
    for B in (1, 2):
        assert ops.gemm_kernel_name(B * S, D, 4 * D, 2, a_packed=True) == "gemm_d8r192_kernel<128, 2>", B
    assert ops.gemm_kernel_name(S, D, D, 2, a_packed=True) == "gemm_d8r192_kernel<128, 2>"
    # a 192-row tiling that would run past the packed row slots (ceil(M / 256) * 256) is not offered: M = 500 -> 3 x 192 = 576 > 512
    assert ops.gemm_kernel_name(500, D, 4 * D, 2, a_packed=True) == "gemm_d8_kernel<128, 2>"
