#!/usr/bin/env python3
"""Record, per rough golden case and refine mode, by how much the order-independent pipeline of this library
(own-trajectory map under methods.neargrid's stepping rule, basins numbered by their smallest voxel, then the
reference's refinement semantics) deviates from the reference's sequential result (tests/golden/r*.npz, produced
by running the reference itself: make_golden.py).  Uses the CPU oracle, which tests/test_oracle_rough.py pins to
those same fixtures first.  Output: tests/golden/rough_expected.json -- asserted by the CPU tests (oracle) and by
the -m gpu tests (HIP path)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from rough_common import ROUGH_CASES, MODES, load_rough, pipeline_maps, deviation  # noqa: E402


def main():
    out = {}
    for name in ROUGH_CASES:
        g, rho = load_rough(name)
        maps = pipeline_maps(g, rho)
        out[name] = {'n_maxima': int(maps['maxima'].shape[0]),
                     'maxima_set_equal': bool(set(maps['maxima'].tolist()) == set(maps['ref_maxima'].tolist())),
                     'maxima_order_equal': bool(np.array_equal(maps['maxima'], maps['ref_maxima'])),
                     'pre_refine_label_diff_vs_ref_main': int((maps['assign'] != g['ng_main']).sum())}
        for tag in MODES:
            if tag not in g.files:
                continue
            out[name][tag] = deviation(g, maps, tag)
        print(name, json.dumps(out[name]))
    with open(os.path.join(HERE, 'rough_expected.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write('\n')


if __name__ == '__main__':
    main()
