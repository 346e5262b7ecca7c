"""CPU: the oracle against the committed regression vectors of tests/golden/ (written by tests/golden/make_golden.py when
the oracle passed the reference's known-answer tests; they are oracle outputs, not reference outputs)."""
import json
import os

import numpy as np

from tests.golden import make_golden


def test_oracle_matches_its_committed_vectors(oracle):
    with open(os.path.join(os.path.dirname(make_golden.__file__), "oracle_regression.json")) as f:
        want = json.load(f)
    got = make_golden.compute()
    assert set(got) == set(want)
    for key, w in want.items():
        g = got[key]
        if isinstance(w, list) and w and isinstance(w[0], float):
            assert np.allclose(g, w, rtol=1e-6, atol=1e-9), key      # float64 sums of fp32 terms: libm-independent, compiler-stable
        else:
            assert g == w, key                                       # integer decisions and hashes of integer images: exact
