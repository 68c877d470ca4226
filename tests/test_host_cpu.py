"""Host-side logic and the C-ABI surface, on CPU (no kernel is launched here)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------ C ABI
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "paced_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(paced_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    import ctypes
    from pacednegatives_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/paced_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signature table and header disagree"
    assert _lib.load().paced_abi_version() == _lib.EXPECTED_ABI == 4


def test_a_library_of_another_abi_version_is_refused(tmp_path, monkeypatch):
    """Argument lists changed between ABI versions: a build that exports every symbol but reports another version must be
    refused at load, not called with shifted arguments."""
    import subprocess
    from pacednegatives_amd import _lib
    names = declared_symbols()
    src = tmp_path / "old.c"
    src.write_text("".join(f"int {n}(void) {{ return 0; }}\n" for n in names if n != "paced_abi_version") + "int paced_abi_version(void) { return 3; }\n")
    so = tmp_path / "libold.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
    monkeypatch.setattr(_lib, "LIB_PATH", str(so))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.PacedHipError, match="ABI version 3.*expects 4"):
        _lib.load()


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _dn, fn in os.walk(os.path.join(ROOT, "pacednegatives_amd")):
        for f in fn:
            if f.endswith(".py") and f != "selftest.py":
                if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dp, f)).read(), flags=re.M):
                    bad.append(f)
    for f in ("train/train_lce.py",):
        if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, f)).read(), flags=re.M):
            bad.append(f)
    assert not bad, bad


# ------------------------------------------------------------------------------------ dataset / sampler
class Corpus:
    def __init__(self, n_docs, n_queries):
        self.d = [{"doc_id": f"d{i}", "text": f"document text number {i}"} for i in range(n_docs)]
        self.q = [{"query_id": f"q{i}", "text": f"what is query {i}"} for i in range(n_queries)]

    def docs_iter(self):
        return iter(self.d)

    def queries_iter(self):
        return iter(self.q)


def make_ds(n_neg, n, var=0.01):
    from pacednegatives_amd.dataloader import LCEDataset
    pairs = [[f"q{i}", f"d{n_neg + i}"] for i in range(3)]
    neg_idx = [[f"d{j}" for j in range(n_neg)] for _ in range(3)]
    return LCEDataset(pairs, neg_idx, Corpus(n_neg + 4, 3), None, 0.5, 4, var, n, 1e-10, 1.0 - 1e-10)


def test_lcedataset_sample_streams_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "sampler.npz"))
    for n_neg in (30, 1000):
        for n in (2, 8, 16):
            ds = make_ds(n_neg, n)
            for wi, w in enumerate(g["weights"]):
                for seed in (0, 1, 2):
                    np.random.seed(seed)
                    got = np.stack([ds.sample(float(w)) for _ in range(4)])
                    assert np.array_equal(got, g[f"nneg{n_neg}_n{n}_w{wi}_s{seed}"]), (n_neg, n, wi, seed)


def test_lcedataset_items_and_failure(golden_dir):
    items = json.load(open(os.path.join(golden_dir, "items.json")))
    ds = make_ds(30, 2)
    np.random.seed(5)
    ds.weight = 0.4
    assert ds[0] == items["item0"]
    assert ds[2] == items["item2"]
    assert len(ds) == items["len"]
    from pacednegatives_amd.dataloader import LCEDataset
    big = LCEDataset([["q0", "d1000"]], [[f"d{j}" for j in range(1000)]], Corpus(1004, 1), None, 0.5, 4, 0.01, 41, 1e-10, 1 - 1e-10)
    with pytest.raises(ValueError, match="Failed to sample"):
        big.sample(1e-10)


def test_legacy_loaders_pick_by_weight():
    from pacednegatives_amd.dataloader import LevelLoader, PairLoader, TripletDataset
    n_neg = 11
    td = TripletDataset([[f"q{i}", f"d{20 + i}"] for i in range(4)], [[f"d{j}" for j in range(n_neg)]] * 4, Corpus(30, 4))
    assert td.n_neg == n_neg - 1
    q, p, n = td.get_items(1, 0.55)
    assert n == "document text number 5"            # floor(0.55 * 10)
    px, nx, op, on = LevelLoader(td, 2).get_batch(1, 1.0)
    assert len(px) == 2 and op == ["true", "true"] and on == ["false", "false"]
    assert nx[0].endswith("Document: document text number 10 Relevant:")
    px, nx, _, _ = PairLoader(td, 2).get_batch(0, torch.tensor([0.0, 0.3]))
    assert nx[1].endswith("number 3 Relevant:")


# ------------------------------------------------------------------------------------ schedules / params
def test_linear_schedule_matches_oracle():
    from oracle.lce_ref import linear_schedule
    from pacednegatives_amd.lceT5 import LinearWarmupSchedule

    class Opt:
        base_lr = 2e-3
        param_groups = [{"lr": 2e-3}]
    for warm, total in ((0, 10), (3, 10), (625, 6250)):
        o = Opt()
        o.param_groups = [{"lr": 2e-3}]
        s = LinearWarmupSchedule(o, warm, total)
        for k in range(total + 3):
            assert abs(o.param_groups[0]["lr"] - 2e-3 * linear_schedule(k, warm, total)) < 1e-15
            s.step()


def test_param_layout_covers_hf_names_once():
    from oracle.t5_ref import T5Dims as RefDims, param_names
    from pacednegatives_amd.config import NAMED
    from pacednegatives_amd.params import hf_name_map, layout
    for name in ("t5-tiny", "t5-small", "t5-base"):
        dims = NAMED[name]
        dims.validate()
        rd = RefDims(dims.vocab_size, dims.d_model, dims.d_kv, dims.num_heads, dims.d_ff, dims.num_layers, dims.num_decoder_layers)
        want = dict(param_names(rd))
        lay = layout(dims)
        covered = {}
        for hf, seg, r0, rc in hf_name_map(dims):
            assert hf in want and hf not in covered
            shape = lay[seg][1]
            cols = shape[1:] if len(shape) > 1 else ()
            assert (rc,) + tuple(cols) == tuple(want[hf]), hf
            covered[hf] = (seg, r0, rc)
        assert set(covered) == set(want)
        # fused segments are tiled exactly by their parts
        rows = {}
        for seg, r0, rc in covered.values():
            rows.setdefault(seg, []).append((r0, rc))
        for seg, parts in rows.items():
            parts.sort()
            assert parts[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert parts[-1][0] + parts[-1][1] == lay[seg][1][0]
        offs = [v[0] for k, v in lay.items() if k != "__total__"]
        assert all(o % 64 == 0 for o in offs)
    assert sum(np.prod(s) for s in dict(param_names(RefDims())).values()) == 222_903_552   # t5-base


def test_relative_position_buckets_match_oracle():
    from oracle.t5_ref import relative_position_bucket as ref
    from pacednegatives_amd.t5_engine import relative_position_bucket as got
    rel = torch.arange(-600, 601)
    for bi in (True, False):
        assert torch.equal(got(rel, bi, 32, 128), ref(rel, bi, 32, 128))


def test_config_round_trip(tmp_path):
    from pacednegatives_amd.config import NAMED, T5Dims, resolve
    NAMED["t5-base"].save(str(tmp_path))
    assert resolve(str(tmp_path)) == NAMED["t5-base"]
    with pytest.raises(ValueError):
        T5Dims(d_kv=32).validate()
    with pytest.raises(ValueError):
        resolve("t5-nonexistent")


def test_synthetic_tokenizer_and_collate_layout():
    from pacednegatives_amd.lceT5 import LCEDataModule
    from pacednegatives_amd.tokenizer import SyntheticTokenizer
    tok = SyntheticTokenizer(32128)
    assert tok(["true"])["input_ids"].tolist() == [[1176, 1]]
    assert tok(["false", "false"]).input_ids.tolist() == [[6136, 1]] * 2
    dm = LCEDataModule("unused", Corpus(4, 1), tok, batch_size=2, n=2)
    batch = [{"pos": "Query: a b Document: c Relevant:", "neg": ["Query: a b Document: d e f Relevant:", "Query: a b Document: g Relevant:"]}] * 2
    p, n = dm.collate()(batch)
    assert p["input_ids"].shape[0] == 2 and n["input_ids"].shape[0] == 4
    assert p["labels"].tolist() == [[1176, 1]] * 2 and n["labels"].tolist() == [[6136, 1]] * 4
    assert n["input_ids"].shape[1] > p["input_ids"].shape[1]          # padded separately
    assert (n["input_ids"][1, -2:] == 0).all()                        # pad id 0, no mask returned
    assert set(p) == {"input_ids", "labels"}


def test_train_cli_flag_surface():
    """Flag names and defaults of the reference's main() (train/train_lce.py:20-37)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train", "train_lce.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--data", "--dataset_name", "--out_dir", "--total_steps", "--eta", "--batch_size", "--lr", "--meta_lr", "--var", "--n",
                 "--use_max", "--warmup_steps", "--sample", "--use_mean", "--num_gpus", "--wandb_project", "--num_workers", "--max_steps"):
        assert flag in out.stdout, flag
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_lce", os.path.join(ROOT, "train", "train_lce.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import inspect
    sig = inspect.signature(mod.main)
    want = {"total_steps": 100000, "eta": 0.0, "batch_size": 16, "lr": 0.001, "meta_lr": None, "var": 0.01, "n": 2, "use_max": False,
            "warmup_steps": 10000, "sample": False, "use_mean": False, "num_gpus": 1, "wandb_project": None, "num_workers": 8,
            "max_steps": None}
    for k, v in want.items():
        assert sig.parameters[k].default == v, k
    assert list(sig.parameters)[:3] == ["data", "dataset_name", "out_dir"]


def test_mining_record_contract():
    from pacednegatives_amd.mining import easy_to_hard, group_results
    rows = torch.tensor([[5, 3, 9], [1, 2, 0]])
    assert easy_to_hard(rows).tolist() == [[9, 3, 5], [0, 2, 1]]
    res = [{"qid": 7, "doc_id_b": ["a", "b"]}]
    recs = group_results(res, [{"query_id": "7", "doc_id_a": "p1"}, {"query_id": "8", "doc_id_a": "p2"}])
    assert recs == [{"query_id": "7", "doc_id_a": "p1", "doc_id_b": ["a", "b"]}]


def test_missing_library_fails_loudly(monkeypatch):
    from pacednegatives_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpaced_hip.so")
    with pytest.raises(_lib.PacedHipError, match="no CPU fallback"):
        _lib.load()


def test_reference_import_line_resolves_to_this_package():
    """train/train_lce.py:4 of the reference, unedited, and the other public modules under the reference's package name."""
    from pacednegatives.lceT5 import LCEModel, LCEDataModule, ChangeDifficulty
    import pacednegatives.dataloader as dl
    import pacednegatives.weights as wt
    import pacednegatives.pairwrapper as pw
    import pacednegatives_amd.lceT5 as impl
    assert LCEModel is impl.LCEModel and LCEDataModule is impl.LCEDataModule and ChangeDifficulty is impl.ChangeDifficulty
    for name in ("LCEDataset", "TripletDataset", "PairLoader", "LevelLoader", "TripletLoader"):
        assert hasattr(dl, name), name
    for name in ("EtaWeights", "LCEWeights", "Weights"):
        assert hasattr(wt, name), name
    for name in ("PacedWrapper", "StdWrapper", "NewWrapper", "adjust_lr"):
        assert hasattr(pw, name), name


def test_setup_reads_records_as_the_reference_does_with_pandas(tmp_path):
    """LCEDataModule.setup reads the records with json; the reference reads them with
    pd.read_json(f, orient='records', dtype={'query_id': str, 'doc_id_a': str, 'doc_id_b': List[str]}) (lceT5.py:57-62).
    Same pairs and negative lists for ids with leading zeros, numeric strings, JSON integers and plain strings."""
    import json
    from typing import List
    pd = pytest.importorskip("pandas")
    from pacednegatives_amd.lceT5 import LCEDataModule
    from pacednegatives_amd.synthetic import ListCorpus
    recs = [{"query_id": "007", "doc_id_a": "0012", "doc_id_b": ["003", "14", "0005"]},
            {"query_id": "1048585", "doc_id_a": "7187158", "doc_id_b": ["7187157", "99", "100"]},
            {"query_id": 12, "doc_id_a": 34, "doc_id_b": [5, 6, 7]},
            {"query_id": "q9", "doc_id_a": "d1", "doc_id_b": ["d2", "d3", "d4"]}]
    path = tmp_path / "recs.json"
    path.write_text(json.dumps(recs))
    with open(path) as f:
        ref = pd.read_json(f, orient="records", dtype={"query_id": str, "doc_id_a": str, "doc_id_b": List[str]})
    want_pairs = ref[["query_id", "doc_id_a"]].values.tolist()
    want_negs = [list(x) for x in ref["doc_id_b"].values]
    ids = {x for r in recs for x in r["doc_id_b"]} | {str(r["doc_id_a"]) for r in recs}
    corpus = ListCorpus([{"doc_id": i, "text": f"doc {i}"} for i in ids],
                        [{"query_id": str(r["query_id"]), "text": f"query {r['query_id']}"} for r in recs])
    dm = LCEDataModule(str(path), corpus, tokenizer=None, batch_size=2, n=2)
    dm.setup()
    assert dm.pairs == want_pairs
    assert dm.neg_idx == want_negs and [type(x[0]) for x in dm.neg_idx] == [type(x[0]) for x in want_negs]
    item = dm.dataset[2]            # the record with integer ids resolves against a corpus keyed the same way
    assert item["pos"] == "Query: query 12 Document: doc 34 Relevant:" and len(item["neg"]) == 2
