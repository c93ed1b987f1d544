"""GPU parity against OUTPUTS OF THE REFERENCE ITSELF (-m gpu): the HIP path, through the C-ABI, re-runs the recipes
behind the two real-data dumps the reference repository holds for its 11 `data/tiny` scans (fixture
tests/golden/tiny_track.npz; the CPU twin of this file is test_oracle_reference_dump.py):

  * img/dead_reckoning/tiny_10.npz : warp (warp.hip) -> DoH maxima (doh.hip) + blob_doh bookkeeping -> pyramid + LK
    (pyrklt.hip) for ten frames must reproduce all 257 saved feature rows, 232 of them bit for bit;
  * img/blob/tiny/*.jpg            : blob_doh and adaptiveNMS (DoH, skimage-order pruning, numpy-1.22 tie order, SSC on
    the device) must land on the circles the reference drew."""
import numpy as np
import pytest

import oracle
from test_oracle_reference_dump import FIX, W, check_against_tiny10, legacy_track, overlay_agreement

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from radarslampy_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def fix():
    return np.load(FIX)


@pytest.fixture(scope="module")
def warped(ctx, fix):
    """float32 and quantised u8 Cartesian images of the 11 scans from the device warp (payload-only records)"""
    out = []
    for p in fix["payload"]:
        f32, u8 = ctx.polar_to_cart_record_u8(np.ascontiguousarray(p), payload_off=0, clip=p.shape[1], want_f32=True, want_u8=True)
        out.append((f32, u8))
    return out


def test_tiny10_feature_dump_reproduced_by_the_hip_path(ctx, fix, warped):
    from radarslampy_amd import getFeatures as gf

    def detect(i):
        blobs = gf.getBlobsFromCart(warped[i][0], min_sigma=0.01, max_sigma=10, num_sigma=3, threshold=.0005)
        return np.fliplr(blobs[:, :2])

    def klt(a, b, pts):
        return ctx.klt_track(warped[a][1], warped[b][1], pts)

    ours = legacy_track(None, detect, klt)
    check_against_tiny10(ours, fix["blobCoord_ref"])
    # and it is the oracle's result, bit for bit
    carts = [oracle.convertPolarImageToCartesian(p.astype(np.float32) / np.float32(255.)) for p in fix["payload"]]
    pyr = [oracle.build_pyramid(oracle.quantize_u8(c), 3) for c in carts]
    want = legacy_track(None, lambda i: np.fliplr(oracle.blob_doh(np.asarray(carts[i], np.float64), min_sigma=0.01, max_sigma=10,
                                                                  num_sigma=3, threshold=.0005)[:, :2]),
                        lambda a, b, pts: oracle.klt_on_pyramids(pyr[a], pyr[b], pts))
    assert np.array_equal(ours, want)


def test_blob_and_anms_circles_by_the_hip_path(ctx, fix, warped):
    from radarslampy_amd import getFeatures as gf
    tot = np.zeros(5, int)
    exact_frames = 0
    for f in range(11):
        blobs = gf.getBlobsFromCart(warped[f][0], min_sigma=0.01, max_sigma=10, num_sigma=3, threshold=.0005)
        sel = gf.adaptiveNMS(warped[f][0], blobs)                       # device SSC
        r = np.array(overlay_agreement(fix, f, blobs, sel))
        assert r[0] <= 2 and r[1] <= 2, (f, r)
        tot += r
        exact_frames += (r[3] == 0 and r[4] == 0)
    assert tot[0] <= 4 and tot[1] <= 6, tot
    assert tot[2] >= 0.985 * (tot[2] + tot[3]) and tot[4] <= 0.015 * (tot[2] + tot[4]), tot
    assert exact_frames >= 5
