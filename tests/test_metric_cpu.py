"""CPU: depthinspace_amd/co/metric.py (device-side DistanceMetric / OutlierFractionMetric of test_epoch / retest) equals the
reference's numpy metrics (co/metric.py:104-154) on the fixture oracle/make_golden.py generated from the imported reference."""
import os
import numpy as np
import torch


def test_metrics_match_reference(golden_dir):
    from depthinspace_amd.co import metric as M
    G = np.load(os.path.join(golden_dir, 'ops.npz'))
    met = M.MultipleMetric(M.DistanceMetric(vec_length=1),
                           M.OutlierFractionMetric(vec_length=1, thresholds=[0.1, 0.5, 1, 2, 5]))
    for k in range(3):
        met.add(torch.from_numpy(G[f'met_es{k}']), torch.from_numpy(G[f'met_gt{k}']))
    vals = met.get()
    assert list(vals.keys()) == [str(k) for k in G['met_keys']]
    for k, ref in zip(G['met_keys'], G['met_vals']):
        tol = 0.0 if str(k).startswith('of') else 2e-6   # counts are exact; moments differ by float32 vs float64 sums
        assert abs(vals[str(k)] - float(ref)) <= tol * max(1.0, abs(float(ref))), (k, vals[str(k)], float(ref))
    assert 'dist2_mean=' in str(met) and 'of0.1=' in str(met)
