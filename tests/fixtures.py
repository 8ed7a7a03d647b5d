"""The committed fixture corpus (tests/golden).  Every test module that parametrises over files takes its list from here:
an empty glob is an error at collection time, never a silently skipped `[NOTSET]` case (round 2's stream-JPEG parity test
globbed the wrong directory and was skipped on every run without anybody noticing)."""
import glob
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _must(paths, what, at_least):
    assert len(paths) >= at_least, f"fixture corpus incomplete: {len(paths)} {what} found under {GOLDEN}, expected >= {at_least}"
    for p in paths:
        assert os.path.getsize(p) > 0, p
    return paths


def jpegs(pattern="*.jpg", issue35=True, at_least=8):
    """generated JPEGs (tools/make_fixtures.py) + the reference's own issue35.jpg"""
    paths = sorted(glob.glob(os.path.join(GOLDEN, "jpeg", pattern)))
    if issue35:
        paths.append(os.path.join(GOLDEN, "ref_images", "issue35.jpg"))
    return _must(paths, f"JPEG files matching {pattern}", at_least)


def ref_pngs(at_least=7):
    """the reference's PNG fixtures (test-images/)"""
    return _must(sorted(glob.glob(os.path.join(GOLDEN, "ref_images", "*.png"))), "reference PNG files", at_least)


def ref_image(name):
    p = os.path.join(GOLDEN, "ref_images", name)
    assert os.path.getsize(p) > 0, p
    return p
