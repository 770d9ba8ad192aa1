"""Host-side tools: the deal notation of tools/deal_search.py must always describe a complete schedule of the fused
kernel (greb_member.hip: 3 ST, 3 FT, 1 S1 and 5 F1 passes, each exactly once, nothing on the polar wave), or a variant
library would silently skip or double rows."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("deal_search", os.path.join(ROOT, "tools", "deal_search.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_candidate_deal_is_a_complete_schedule():
    ds = _load()
    want = sorted([("kST", i) for i in range(3)] + [("kFT", i) for i in range(3)] + [("kS1", 0)] + [("kF1", i) for i in range(5)])
    assert ds.DEALS, "no candidate deals"
    for name, deal in ds.DEALS.items():
        rows = re.findall(r"\{(\{[^{}]*\}(?:,\{[^{}]*\}){2})\}", ds.table(deal))
        assert len(rows) == 8, (name, len(rows))
        cells = [re.findall(r"\{(k\w+),(\d+)\}", r) for r in rows]
        assert all(len(c) == 3 for c in cells), name
        pw = ds.polar_wave(deal)
        assert all(k == "kNone" for k, _ in cells[pw]), (name, "work dealt to the polar wave")
        got = sorted((k, int(i)) for w, c in enumerate(cells) for k, i in c if k != "kNone")
        assert got == want, (name, got)


def test_shipped_deal_matches_a_candidate():
    """The table compiled into the release library is one of the measured candidates (the first of the list)."""
    ds = _load()
    src = open(os.path.join(ROOT, "greb_climate_model_amd", "csrc", "greb_member.hip")).read()
    m = re.search(r"#else\s+constexpr Pass t\[8\]\[3\] = \{(.*?)\};\s+#endif", src, re.S)
    assert m, "deal_fast table not found"
    body = re.sub(r"/\*.*?\*/", "", m.group(1))
    shipped = [re.findall(r"\{(k\w+), (\d+)\}|(none)", r) for r in re.findall(r"\{((?:\{k\w+, \d+\}|none)(?:, (?:\{k\w+, \d+\}|none)){2})\}", body)]
    norm = [[("kNone", 0) if n else (k, int(i)) for k, i, n in row] for row in shipped]
    first = next(iter(ds.DEALS.values()))
    rows = re.findall(r"\{(\{[^{}]*\}(?:,\{[^{}]*\}){2})\}", ds.table(first))
    cand = [[(k, int(i)) for k, i in re.findall(r"\{(k\w+),(\d+)\}", r)] for r in rows]
    assert norm == cand
    assert re.search(r"constexpr int kPolarWaveFast = (\d+);\s+#endif", src).group(1) == str(ds.polar_wave(first))


def test_latency_groups_of_config5():
    """ensemble.max_chain_sweeps restates the reference's sub-cycle count (src/greb.f90:652-653; SURVEY.md App. B: 8 at
    96x48, 225 at 384x192, 1 800 once kappa < 7.27e5 there); latency_groups puts exactly config 5's long-chain members
    into an engine of their own and leaves ensembles without such members in one piece."""
    import sys
    sys.path.insert(0, ROOT)
    from greb_climate_model_amd import abi, ensemble
    assert ensemble.max_chain_sweeps(8e5, 96, 48) == 8
    assert ensemble.max_chain_sweeps(8e5, 384, 192) == 225
    assert ensemble.max_chain_sweeps(7.2e5, 384, 192) == 1800 and ensemble.max_chain_sweeps(7.3e5, 384, 192) == 225
    ov = ensemble.perturbed_physics(64, abi.default_params())
    groups = ensemble.latency_groups(ov[:, 3], 384, 192)
    assert [len(g) for g in groups] == [62, 2]
    assert sorted(groups[1]) == sorted(i for i in range(64) if ov[i, 3] < 7.27e5)
    assert sorted(list(groups[0]) + list(groups[1])) == list(range(64))
    one = ensemble.latency_groups(ov[:, 3], 96, 48)
    assert len(one) == 1 and list(one[0]) == list(range(64))
    assert ensemble.run_beside([lambda: 1, lambda: 2]) == [1, 2]
