"""GPU tests (-m gpu) added in round 3: the T0 known answer through the product, threads at the boundary, the drop-in
shim's device-side [col][row] layout, per-stripe host assembly of groups, the segmented (wait-free) compaction."""
import numpy as np
import pytest

from conftest import golden_calibration, pkg
from test_oracle import golden_relative_geometry, relative_from_projection

pytestmark = pytest.mark.gpu


def test_T0_product_matches_reference_held_opencv_output():
    """The product's host T0 (sl3d_set_calibration: Rodrigues + K[R|t], 7/triangulation.cpp:1069-1116) against the reference-held
    known answer (Triangulation/Relative_geometry/*.xml = OpenCV 2.4's own cvRodrigues2 / cvTranspose / cvGEMM / cvSub output on
    the rotation and translation vectors stage 7 reads, 6/system_calibration.cpp:1488-1516).  With K = I the product's A is
    [R|t]; Rc*Rp^T and tc - (Rc*Rp^T)*tp formed from it in cvGEMM's summation order must equal the 12 stored doubles to <= 1 ulp
    (observed: bit for bit), and the product's A = K[R|t] with the real intrinsics must equal the pinned oracle's."""
    from oracle.oracle import Oracle
    S = pkg("scanner")
    ct, dims = golden_calibration()
    cal = dict(zip(("Kc", "dc", "rc", "tc", "Kp", "dp", "rp", "tp"), ct))
    kat_R, kat_t = golden_relative_geometry()
    I3, z5 = np.eye(3).ravel(), np.zeros(5)
    with S.Scanner(64, 32, dims["PW"], dims["PH"], dims["N_v"], dims["N_h"], dims["fw_v"], dims["fw_h"]) as sc:
        sc.set_calibration(I3, z5, cal["rc"], cal["tc"], I3, z5, cal["rp"], cal["tp"])
        A_cam, A_proj = sc.projection_matrices()
        R, t = relative_from_projection(A_cam, A_proj)
        ulp_R = np.abs(R.view(np.int64) - kat_R.view(np.int64)).max()
        ulp_t = np.abs(t.view(np.int64) - kat_t.view(np.int64)).max()
        assert ulp_R <= 1 and ulp_t <= 1, (ulp_R, ulp_t)
        # the real intrinsics: A = K [R|t] equals the oracle's (whose R is pinned by the same known answer)
        sc.set_calibration(*ct)
        A_cam, A_proj = sc.projection_matrices()
    o = Oracle(16, 16, dims["PW"], dims["PH"], dims["N_v"], dims["N_h"], dims["fw_v"], dims["fw_h"])
    o.set_calibration(*ct)
    oc, op = (np.array(a).reshape(3, 4) for a in o.projection_matrices())
    assert np.array_equal(A_cam, oc) and np.array_equal(A_proj, op)
