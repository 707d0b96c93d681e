"""The generated Student-t tables (product copy and oracle copy) against scipy."""
import os
import re

import numpy as np
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_tables(path):
    text = open(path).read()
    body = text[text.index("_tables["):]
    rows = re.findall(r"\{ /\* alpha = ([0-9.]+) (two|one)-sided.*?\*/(.*?)\},", body, flags=re.S)
    return {(float(a), side): np.array([float(x.rstrip("f")) for x in re.findall(r"[-0-9.e+]+f", vals)], np.float32)
            for a, side, vals in rows}


def test_tables_match_scipy_and_each_other():
    prod = parse_tables(os.path.join(ROOT, "statmc_amd", "csrc", "t_quantiles.h"))
    orac = parse_tables(os.path.join(ROOT, "oracle", "t_quantiles_oracle.h"))
    # table order: two-sided 0.005, 0.002, 0.05, then the one-sided three (statmc_filter_spec.sides)
    assert list(prod) == [(0.005, "two"), (0.002, "two"), (0.05, "two"), (0.005, "one"), (0.002, "one"), (0.05, "one")]
    for (alpha, side), tab in prod.items():
        tails = 2.0 if side == "two" else 1.0
        assert tab.shape == (4096,)
        assert np.array_equal(tab, orac[(alpha, side)])
        ref = stats.t.ppf(1 - alpha / tails, np.arange(1, 4097)).astype(np.float32)
        assert np.array_equal(tab, ref)
        assert (np.diff(tab) <= 0).all() and tab[0] > tab[-1]   # decreasing towards the normal quantile (fp32 ties far out)
        assert abs(tab[-1] - stats.norm.ppf(1 - alpha / tails)) < 2e-3


def test_oracle_lookup(oracle):
    assert np.isinf(oracle.t_quantile(0, 0)) and np.isinf(oracle.t_quantile(0, -3))
    assert np.isclose(oracle.t_quantile(0, 1), 127.3213, rtol=1e-6)
    assert oracle.t_quantile(0, 4096) == oracle.t_quantile(0, 10 ** 6)   # clamped beyond the table
    assert oracle.t_quantile(2, 30) < oracle.t_quantile(0, 30) < oracle.t_quantile(1, 30)
    assert oracle.t_quantile(3, 30) < oracle.t_quantile(0, 30)            # one-sided 0.005 < two-sided 0.005
