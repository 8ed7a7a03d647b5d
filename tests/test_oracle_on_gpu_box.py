"""The checker is checked where it is used: these `gpu`-marked tests re-run the oracle's own pinning on the MI355X box (the
driver runs only `-m gpu` there, so the CPU-marked pinning suite never saw the liboracle.so that judges the kernels on that box).
No kernel is launched here; /root/reference is not needed and no reference-derived code runs: only fixtures (data) are read."""
import pytest

import test_oracle_pinning as P

pytestmark = pytest.mark.gpu


def test_oracle_convert_equals_reference_vectors_on_this_box():
    P.check_convert_against_reference_vectors()


@pytest.mark.parametrize("path", P.JPEGS[:6] + P.JPEGS[-1:], ids=lambda p: p.split("/")[-1])
def test_oracle_jpeg_golden_on_this_box(path):
    P.test_jpeg_golden(path)


def test_oracle_jpeg_reference_derived_vectors_on_this_box():
    """the H2V2 leg: oracle_jpeg.c == tests/golden/jpeg_h2v2_ref.npz (made in the build container by the reference-derived
    restatement; only the data travels)"""
    P.check_oracle_against_jpeg_vectors()


def test_oracle_png_known_answers_on_this_box():
    P.test_png_issue76_known_answer()
    P.test_png_inflate_known_answer()
