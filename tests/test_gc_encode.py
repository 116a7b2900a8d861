"""Native gc row encoder (feos_torch_amd/csrc_host/gc_encode.cpp) against the pure-Python
specification `encode_rows_py` and against the oracle's own encoder.  CPU only."""
import os

import numpy as np
import torch
import pytest

from conftest import ROOT
from feos_torch_amd.gc_pcsaft import encode_rows, encode_rows_py
from feos_torch_amd.synthetic import gc_batch, gc_molecule_library, load_segment_table


@pytest.fixture(scope="module")
def table():
    return load_segment_table(os.path.join(ROOT, "tests", "data", "sauer2014_hetero.json"))


def test_reference_test_molecules(table):
    # the molecule pairs of the reference's tests/test_gc_pcsaft.py:17-42
    ident = [s for s, _ in table]
    segment_lists = [
        [["CH3", "CH2", "CH2", "CH3"], ["CH3", "CH2", "CH3"]],
        [["CH3", ">CH", "CH3", "CH3"], ["CH3", ">C<", "CH3", "CH3", "CH3"]],
        [["CH3", ">CH", "CH3", "CH=O"], ["CH3", ">C<", "CH3", "CH3", "HCOO"]],
        [["CH3", ">CH", "CH=O", "OH"], ["CH3", ">C<", "CH3", "HCOO", "NH2"]],
    ]
    bond_lists = [
        [[[0, 1], [1, 2], [2, 3]], [[0, 1], [1, 2]]],
        [[[0, 1], [1, 2], [1, 3]], [[0, 1], [1, 2], [1, 3], [1, 4]]],
        [[[0, 1], [1, 2], [1, 3]], [[0, 1], [1, 2], [1, 3], [1, 4]]],
        [[[0, 1], [1, 2], [1, 3]], [[0, 1], [1, 2], [1, 3], [1, 4]]],
    ]
    a = encode_rows(ident, segment_lists, bond_lists)
    b = encode_rows_py(ident, segment_lists, bond_lists)
    assert a.dtype == np.uint8 and a.shape == (4, 80)
    assert np.array_equal(a, b)
    # butane / propane by hand: CH3 x2 + CH2 x2, bonds CH2-CH3 x2 + CH2-CH2 x1
    i3, i2 = ident.index("CH3"), ident.index("CH2")
    lo, hi = min(i3, i2), max(i3, i2)
    assert list(a[0, 0:2]) == [lo, hi] and list(a[0, 16:18]) == [2, 2]
    assert list(a[0, 8:10]) == [lo, hi] and list(a[0, 24:26]) == ([2, 1] if lo == i3 else [1, 2])


def test_random_batch_matches_specification(table):
    ident = [s for s, _ in table]
    b = gc_batch(5000, table)
    assert np.array_equal(encode_rows(ident, b["segment_lists"], b["bond_lists"]),
                          encode_rows_py(ident, b["segment_lists"], b["bond_lists"]))
    # the same molecules as fresh objects (tuples instead of lists, no shared identity)
    sl = [[tuple(m0), tuple(m1)] for m0, m1 in b["segment_lists"][:500]]
    bl = [[tuple(tuple(x) for x in m0), [list(x) for x in m1]] for m0, m1 in b["bond_lists"][:500]]
    assert np.array_equal(encode_rows(ident, sl, bl), encode_rows_py(ident, b["segment_lists"][:500], b["bond_lists"][:500]))


def test_matches_oracle_encoding(oracle, table):
    # decode the 80-byte rows back to the dense [n,2,S] counts / [n,2,S,S] bond matrices the reference
    # builds (feos_torch/gc_pcsaft.py:24-63) and compare with the oracle's independent dense encoder
    ident = [s for s, _ in table]
    S = len(ident)
    lib = gc_molecule_library()
    sl = [[lib[i][1], lib[(i + 3) % len(lib)][1]] for i in range(len(lib))]
    bl = [[lib[i][2], lib[(i + 3) % len(lib)][2]] for i in range(len(lib))]
    rows = encode_rows(ident, sl, bl)
    enc = oracle.gc_encode(table, sl, bl, [])
    counts = np.zeros((len(lib), 2, S))
    bonds = np.zeros((len(lib), 2, S, S))
    for r in range(len(lib)):
        for c in range(2):
            for k in range(8):
                if rows[r, 16 + 8 * c + k]:
                    counts[r, c, rows[r, 8 * c + k]] += rows[r, 16 + 8 * c + k]
                if rows[r, 64 + 8 * c + k]:
                    bonds[r, c, rows[r, 32 + 8 * c + k], rows[r, 48 + 8 * c + k]] += rows[r, 64 + 8 * c + k]
    assert np.array_equal(counts, enc["counts"])
    assert np.array_equal(bonds, enc["bonds"])


def test_edge_cases(table):
    ident = [s for s, _ in table]
    assert encode_rows(ident, [], []).shape == (0, 80)
    one = encode_rows(ident, [[["CH3"], ["CH3", "CH3"]]], [[[], [[0, 1]]]])
    assert np.array_equal(one, encode_rows_py(ident, [[["CH3"], ["CH3", "CH3"]]], [[[], [[0, 1]]]]))
    with pytest.raises(KeyError):
        encode_rows(ident, [[["CH3", "nope"], ["CH3"]]], [[[[0, 1]], []]])
    with pytest.raises(IndexError):
        encode_rows(ident, [[["CH3", "CH3"], ["CH3"]]], [[[[0, 2]], []]])
    with pytest.raises(ValueError):
        encode_rows(ident, [[["CH3"], ["CH3"]]], [[[]]])  # a row needs two molecules
    nine = ident[:9]
    with pytest.raises(ValueError):
        encode_rows(ident, [[nine, ["CH3"]]], [[[], []]])  # more than 8 distinct segment types
    with pytest.raises(ValueError):
        encode_rows_py(ident, [[nine, ["CH3"]]], [[[], []]])
    with pytest.raises(ValueError):
        encode_rows(ident, [[["CH3"] * 256, ["CH3"]]], [[[], []]])  # multiplicity above 255


def test_class_order_is_a_permutation_sorted_by_class():
    """native.gc_class_order (pure torch, any device): a permutation of the rows, association class x polarity
    non-increasing along it (expensive classes first), ties by bond- and segment-type entry counts."""
    import os

    import numpy as np
    import torch

    from feos_torch_amd import native
    from feos_torch_amd.gc_pcsaft import build_table, encode_rows
    from feos_torch_amd.synthetic import gc_batch, load_segment_table

    table = load_segment_table(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sauer2014_hetero.json"))
    n = 5000
    b = gc_batch(n, table, seed=5)
    ident = [s for s, _ in table]
    rows = torch.from_numpy(encode_rows(ident, b["segment_lists"], b["bond_lists"]))
    seg = torch.tensor(np.stack([v for _, v in table]), dtype=torch.float64)
    tab = build_table(seg, torch.zeros((len(ident), len(ident)), dtype=torch.float64))
    order = native.gc_class_order(tab, len(ident), rows)
    assert order.dtype == torch.int32 and sorted(order.tolist()) == list(range(n))
    segv = seg.numpy()
    ids, cnt = rows[:, 0:16].numpy().astype(int), rows[:, 16:32].numpy().astype(float)
    par = segv[ids]

    def per_mol(v):
        return (cnt * v).reshape(n, 2, 8).sum(axis=2)

    associating = ((per_mol(par[:, :, 4]) * per_mol(par[:, :, 5])) != 0).sum(axis=1)
    self_assoc = ((per_mol(par[:, :, 6]) * per_mol(par[:, :, 7])) != 0).sum(axis=1)
    cls = np.zeros(n, int)
    cls[(associating == 1) & (self_assoc == 1)] = 1
    cls[(associating == 2) & (self_assoc == 1)] = 2
    cls[(associating == 2) & (self_assoc == 2)] = 3
    key = 2 * cls + (per_mol(par[:, :, 3] ** 2) > 0).any(axis=1)
    k = key[order.numpy()]
    assert np.all(k[:-1] >= k[1:])
    assert len(np.unique(key)) >= 3  # the synthetic library really mixes classes


def test_index_encoding_assembles_the_same_rows():
    """encode_indices (distinct molecules once + two int32 per row) + the gather of encode_rows_device give the rows of the
    host encoder byte for byte (cpu device here; the GPU path is the same torch gather)."""
    import os

    from feos_torch_amd.gc_pcsaft import encode_rows, encode_rows_device
    from feos_torch_amd.synthetic import gc_batch, load_segment_table

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = load_segment_table(os.path.join(root, "tests", "data", "sauer2014_hetero.json"))
    b = gc_batch(5000, table, seed=5)
    ident = [s for s, _ in table]
    rows = encode_rows_device(ident, b["segment_lists"], b["bond_lists"], "cpu")
    assert rows.dtype == torch.uint8 and tuple(rows.shape) == (5000, 80)
    assert np.array_equal(rows.numpy(), encode_rows(ident, b["segment_lists"], b["bond_lists"]))
    # fresh list objects per row (no shared identity): still one table entry per distinct structure is not required, only equal rows
    segs = [[list(a), list(c)] for a, c in b["segment_lists"][:200]]
    bonds = [[[list(p) for p in a], [list(p) for p in c]] for a, c in b["bond_lists"][:200]]
    assert np.array_equal(encode_rows_device(ident, segs, bonds, "cpu").numpy(), encode_rows(ident, segs, bonds))
    assert tuple(encode_rows_device(ident, [], [], "cpu").shape) == (0, 80)
