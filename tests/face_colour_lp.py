"""The linear program that ties the face colours of the reference's 2.off / 3.off to voxel
colours (tools/make_off_fixture.py builds the fixture with it, tests/test_mc_off.py shows which
face-colour rule the files admit)."""
import numpy as np


def face_colour_program(vox_of_vertex, f, rule, nvox, time_limit=None):
    """max t  s.t.  |sum_k w_k x[voxel of corner k] - 3 f| <= 1.5 - t for every face,
    0 <= x <= 255: (status, t, x).  rule "second_twice": weights (1, 2, 0) -- the reference's
    face colour; "three_corners": (1, 1, 1).  t > 0: a colouring exists and survives fp32."""
    from scipy.optimize import linprog
    from scipy.sparse import coo_matrix
    A, B, C = vox_of_vertex[0::3], vox_of_vertex[1::3], vox_of_vertex[2::3]
    terms = [(A, 1.0), (B, 2.0)] if rule == "second_twice" else [(A, 1.0), (B, 1.0), (C, 1.0)]
    n = len(f)
    R = np.arange(n)
    rr, cc, vv = [], [], []
    for sgn, off in ((1.0, 0), (-1.0, n)):
        for idx, w in terms:
            rr.append(R + off)
            cc.append(idx)
            vv.append(np.full(n, sgn * w))
        rr.append(R + off)
        cc.append(np.full(n, nvox))
        vv.append(np.ones(n))
    M = coo_matrix((np.concatenate(vv), (np.concatenate(rr), np.concatenate(cc))),
                   shape=(2 * n, nvox + 1)).tocsr()  # (duplicate entries add up: a == b)
    f = f.astype(np.float64)
    b = np.concatenate([1.5 + 3 * f, 1.5 - 3 * f])
    c = np.zeros(nvox + 1)
    c[-1] = -1.0
    opts = {"time_limit": time_limit} if time_limit else {}
    res = linprog(c, A_ub=M, b_ub=b, bounds=[(0, 255)] * nvox + [(None, 1.5)], method="highs",
                  options=opts)
    if res.status != 0:
        return res.status, None, None
    return 0, float(res.x[-1]), res.x[:-1]
