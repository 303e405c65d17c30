"""CPU: the host-side logic of bench.py that decides what the JSON line claims - the Conv3D tie check behind `disp_l1_vs_ref`
and the entry-point table behind `roofline` - on a reference fixture (no GPU, no timing)."""
import os
import numpy as np
import torch

import bench
from oracle import dis_oracle as O
from depthinspace_amd import synth


def _tap_of_fixture(golden_dir, name='mf_64_bs1'):
    Gs = np.load(os.path.join(golden_dir, name + '.npz'))
    H, W, bs = int(Gs['H']), int(Gs['W']), int(Gs['bs'])
    settings = synth.make_settings(H, W, pattern=str(Gs['pattern']))
    batch = synth.make_batch(settings, bs, 4, seed=int(Gs['bseed']))
    params = O.init_params(O.mf_param_shapes(), seed=int(Gs['pseed']))
    ctx = O.StepContext(settings)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    O.CONV3D_TAP = []
    try:
        with torch.no_grad():
            data = O.copy_data(ctx, tb)
            O.mf_net_forward(ctx, params, data, O.read_optical_flow(data, 4))
    finally:
        tap, O.CONV3D_TAP = O.CONV3D_TAP, None
    # the reference's own torch.topk output on the fixture host: (tl, bs, ho, wo, 9)
    sets = [torch.from_numpy(np.asarray(Gs['knn_idx_core'])), torch.from_numpy(np.asarray(Gs['knn_idx_quarter']))]
    return tap, sets


def test_knn_tie_check_accepts_ties_and_rejects_a_wrong_neighbour(golden_dir):
    tap, sets = _tap_of_fixture(golden_dir)
    assert len(tap) >= 8
    base = bench.knn_tie_check(tap, sets)
    assert base['geometries_checked'] == 8 and base['conv3d_rows'] > 0
    assert base['pass'] and base['non_tie_rows'] == 0   # (whatever this host's tie-breaks are, they are ties)

    # (1) swap a selected neighbour of one row for an unselected candidate with the SAME key value: a tie, still a pass
    e = next(x for x in tap if x['target'] == 0 and x['name'].endswith('conv3d_1'))
    key = e['key'].reshape(-1, 36)
    cur = sets[0][0].reshape(-1, 9).long()
    done = False
    for r in range(key.shape[0]):
        sel = set(cur[r].tolist())
        for j in cur[r].tolist():
            same = [c for c in range(36) if c not in sel and float(key[r, c]) == float(key[r, j])]
            if same:
                s2 = [t.clone() for t in sets]
                flat = s2[0][0].reshape(-1, 9)
                flat[r, cur[r].tolist().index(j)] = same[0]
                out = bench.knn_tie_check(tap, s2)
                assert out['rows_whose_set_differs'] == base['rows_whose_set_differs'] + 1
                assert out['pass'] and out['non_tie_rows'] == 0
                done = True
                break
        if done:
            break
    # (a planar synthetic scene has exact ties; if a fixture ever has none the check below still covers the logic)

    # (2) swap a selected neighbour for the candidate with the LARGEST key of the row: a selection error, must be counted
    r = int(torch.argmax(key.max(dim=1).values - key.min(dim=1).values))
    worst = int(torch.argmax(key[r]))
    assert worst not in cur[r].tolist()
    s3 = [t.clone() for t in sets]
    s3[0][0].reshape(-1, 9)[r, 0] = worst
    bad = bench.knn_tie_check(tap, s3)
    assert not bad['pass'] and bad['non_tie_rows'] == 1
    assert bad['largest_relative_key_gap_of_a_differing_row'] > 1e-2


def test_conv3x3_form_table_reads_the_recorded_int_arguments():
    """(n, h, w), padding and fused operands of every entry point that reaches conv_f16x2_kernel, from the int arguments in the
    order depthinspace_amd/ops.py passes them (lib.call records exactly those)."""
    F = bench.CONV3X3_FORMS
    acc = 0x100
    assert F['dis_conv2d_fwd_bf16x3']((16, 128, 108, 32, 32, 3, 1, 1, 1), 5) == {'nhw': (16, 128, 108), 'pad': 1, 'in_extra': 0, 'out_extra': 0}
    assert F['dis_conv2d_fwd_bf16x3_oihw']((1, 32, 32, 288, 16, 128, 108, 32, 32, 3, 1, 1, acc), 4)['out_extra'] == 1
    assert F['dis_conv2d_fwd_bf16x3_oihw']((0, 32, 32, 288, 16, 128, 108, 32, 32, 3, 1, 1, 1), 5)['nhw'] == (16, 128, 108)
    assert F['dis_conv2d_fwd_bf16x3_gn']((32, 32, 288, 16, 128, 108, 32, 32, 3, 1, 1, 0), 7)['nhw'] == (16, 128, 108)
    d = F['dis_conv2d_dgrad_bf16x3_act']((1, 32, 32, 288, 16, 128, 108, 32, 32, 1, 1), 4)
    assert d == {'nhw': (16, 128, 108), 'pad': 1, 'in_extra': 1, 'out_extra': 1}
    assert F['dis_conv2d_dgrad_bf16x3_gnsums']((32, 32, 288, 16, 128, 108, 32, 32, 1), 5)['out_extra'] == 1
    assert F['dis_conv2d_dgrad_bf16x3_gnsums_res']((32, 32, 288, 16, 128, 108, 32, 32, 1), 6)['out_extra'] == 3
    assert F['dis_conv2d_dgrad_bf16x3_gnsums_res']((32, 32, 288, 16, 128, 108, 32, 32, 1), 5)['out_extra'] == 2
    # every form is named in the library's signature table and takes the int count the table above indexes
    from depthinspace_amd import lib
    for name in F:
        assert name in lib.SIGS
    assert lib.SIGS['dis_conv2d_dgrad_bf16x3_act'].count('i') == 11
    assert lib.SIGS['dis_conv2d_dgrad_bf16x3_gnsums'].count('i') == 9
    assert lib.SIGS['dis_conv2d_dgrad_bf16x3_gnsums_res'].count('i') == 9
    assert lib.SIGS['dis_conv2d_fwd_bf16x3_oihw'].count('i') == 13
    assert lib.SIGS['dis_conv2d_fwd_bf16x3_gn'].count('i') == 12
