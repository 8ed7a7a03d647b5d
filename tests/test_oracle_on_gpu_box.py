"""The checker is checked where it is used: these `gpu`-marked tests re-run the oracle's own pinning on the MI355X box (the
driver runs only `-m gpu` there, so the CPU-marked pinning suite never saw the liboracle.so that judges the kernels on that box).
No kernel is launched here; /root/reference is not needed (golden fixtures + tools/ref_literal_jpeg.py travel with the repo)."""
import numpy as np
import pytest

import test_oracle_pinning as P

pytestmark = pytest.mark.gpu


def test_oracle_convert_equals_reference_vectors_on_this_box():
    P.check_convert_against_reference_vectors()


@pytest.mark.parametrize("path", P.JPEGS[:6] + P.JPEGS[-1:], ids=lambda p: p.split("/")[-1])
def test_oracle_jpeg_golden_on_this_box(path):
    P.test_jpeg_golden(path)


def test_oracle_jpeg_literal_restatement_on_this_box():
    """the H2V2 leg: oracle_jpeg.c == tools/ref_literal_jpeg.py bit for bit (a 12 000-block sample of the CPU suite's 10^5)"""
    R = P._literal()
    rng = np.random.default_rng(5)
    for mz in (1, 2, 3, 6, 10, 15, 21, 28, 36, 64):
        blocks = P._pinning_blocks(rng, 1200)
        temps, samples = R.chroma_expand(blocks, mz)
        pix = R.idct(blocks, mz)
        for i in range(0, 1200, 3):
            up = P.O.jpeg_upsample_block(blocks[i], mz)
            assert np.array_equal(up.reshape(4, 64), temps[:, i, :]) and np.array_equal(P.O.jpeg_idct(blocks[i], mz).reshape(64), pix[i])
            for q in range(4):
                assert np.array_equal(P.O.jpeg_idct_4x4(up[q]).reshape(64), samples[q, i])
    co = P._pinning_blocks(rng, 6 * 60).reshape(60, 6, 64)
    assert np.array_equal(P.O.jpeg_reconstruct(160, 96, 3, P.O.JPGD_YH2V2, co, None, 4), R.decode_h2v2_rgba_fast(co, 160, 96))


def test_oracle_png_known_answers_on_this_box():
    P.test_png_issue76_known_answer()
    P.test_png_inflate_known_answer()
