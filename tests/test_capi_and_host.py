"""CPU-side checks: the C-ABI library builds/loads and exports every symbol include/gtos_hip.h declares (no compute
calls without a GPU), the ctypes binding mirrors the header, the product refuses to run off-GPU, and the host logic
(synthetic batches, lr schedule, flat buckets) behaves."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "gtos_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(gtos_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        out[m.group(1)] = [] if args == ["void"] else args
    return out


@pytest.fixture(scope="module")
def lib_path():
    from gtos_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(lib_path):
    funcs = header_functions()
    assert len(funcs) >= 15
    lib = ctypes.CDLL(lib_path)
    for name in funcs:
        assert hasattr(lib, name), "libgtos_hip.so does not export %s" % name
    lib.gtos_abi_version.restype = ctypes.c_int
    from gtos_amd import _lib
    assert lib.gtos_abi_version() == _lib.ABI_VERSION


def test_host_library_exports_every_declared_symbol():
    from gtos_amd import build
    lib = ctypes.CDLL(build.build_host(verbose=False))
    src = open(os.path.join(ROOT, "include", "gtos_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(gtos_(?:relbatch|pathtrie|relindex)_\w+)\s*\(", src)
    assert set(names) == {"gtos_relbatch_build", "gtos_relbatch_dims", "gtos_relbatch_export", "gtos_relbatch_free", "gtos_relbatch_csr",
                          "gtos_pathtrie_build", "gtos_pathtrie_sizes", "gtos_pathtrie_export", "gtos_pathtrie_free",
                          "gtos_pathtrie_derived_sizes", "gtos_pathtrie_export_derived",
                          "gtos_relindex_build", "gtos_relindex_sizes", "gtos_relindex_export", "gtos_relindex_free"}
    for n in names:
        assert hasattr(lib, n)


def test_ctypes_binding_mirrors_header(lib_path):
    from gtos_amd import _lib
    funcs = header_functions()
    assert set(funcs) == set(_lib.SIGNATURES), set(funcs) ^ set(_lib.SIGNATURES)

    def kind(arg):
        if "*" in arg:
            return ctypes.c_void_p
        base = arg.split()[-2] if len(arg.split()) > 1 else arg
        return {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "uint64_t": ctypes.c_uint64,
                "size_t": ctypes.c_size_t}[base]
    for name, args in funcs.items():
        want = [kind(a) for a in args]
        assert want == _lib.SIGNATURES[name], name
    _lib.load()


def test_product_refuses_cpu_tensors(lib_path):
    from gtos_amd import ops, _lib
    with pytest.raises(_lib.GtosHipError):
        ops.gemm(torch.randn(4, 4), torch.randn(4, 4))
    from gtos_amd.graph_transformer import GraphTransformer
    m = GraphTransformer(1, 16, 32, 2, 0.0)
    with pytest.raises(_lib.GtosHipError):
        m(torch.randn(3, 2, 16), torch.randn(3, 3, 2, 16))


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may touch oracle/."""
    for sub_ in ("gtos_amd", "tools"):
        pkg = os.path.join(ROOT, sub_)
        for fn in os.listdir(pkg):
            if fn.endswith(".py"):
                assert "oracle" not in open(os.path.join(pkg, fn)).read(), fn
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import", src)]
    # each import sits inside a cpu-baseline leg: the function cpu_baseline() or the guarded block of decode_bench()
    assert len(uses) == 2
    assert "def cpu_baseline" in src[:uses[0]] and "if not a.no_cpu_baseline:" in src[uses[0]:uses[1]]


def test_state_dict_keys_match_reference_layout():
    from gtos_amd.graph_transformer import GraphTransformer
    from conftest import load_golden, sub
    g = load_golden("gt_pad")
    L, d, ff, H, n, B = [int(v) for v in g["cfg"]]
    m = GraphTransformer(L, d, ff, H, 0.0)
    sd = sub(g, "sd/")
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd)
    assert m.layers[0].self_attn.in_proj_weight.shape == (3 * d, d)
    assert m.layers[0].self_attn.relation_in_proj.weight.shape == (2 * d, d)


def test_generator_loads_reference_checkpoint_layout():
    from gtos_amd.generator import Generator
    from oracle.gtos_oracle import VocabSpec
    from conftest import load_golden, sub
    from tests_support import SMALL_VOCAB, SMALL_GEN_ARGS
    g = load_golden("gen_small")
    d, ff, H, gl = [int(v) for v in g["cfg"]]
    vocabs = {k: VocabSpec(v, 0) for k, v in SMALL_VOCAB.items()}
    m = Generator(vocabs, *SMALL_GEN_ARGS, d, ff, H, 0.0, 1, gl, 2, None, torch.device("cpu"))
    missing, unexpected = m.load_state_dict(sub(g, "sd/"), strict=True)
    assert not missing and not unexpected


# ---------------------------------------------------------------------------------------------- synthetic batches
def test_synth_batch_layout_and_determinism():
    from gtos_amd import synth
    b1, s1 = synth.make_config_batch("C1")
    b2, s2 = synth.make_config_batch("C1")
    assert s1 == s2 and all(torch.equal(b1[k], b2[k]) for k in b1)
    n, B = s1["n"], s1["B"]
    assert (n, B) == (21, 8)
    rel = b1["relation"]
    assert rel.shape == (n, n, B) and rel.dtype == torch.int64
    # <CLS> conventions of generator/data.py:138-147
    assert int(rel[0, 0, 0]) == 2 and bool((rel[1:, 0, :] == 0).all()) and bool((rel[0, 1:, :] == 1).all())
    bank, blen = b1["relation_bank"], b1["relation_length"]
    R = s1["R"]
    assert bank.shape[1] == R and blen.shape == (R,) and int(rel.max()) == R - 1
    assert int(blen.max()) <= 8 and int(blen.min()) >= 1
    assert bank[:, 0].tolist()[:1] == [synth.REL_CLS] and int(bank[0, 1]) == synth.REL_RCLS and int(bank[0, 2]) == synth.REL_SELF
    for r in range(0, R, max(1, R // 50)):              # padding past the length is 0
        assert bool((bank[int(blen[r]):, r] == 0).all()) and bool((bank[:int(blen[r]), r] != 0).all())
    assert bool((torch.diagonal(rel[1:, 1:, 0]) == 2).all())      # self paths
    assert b1["concept"].shape == (n, B) and bool((b1["concept"][0] == synth.CONCEPT_CLS).all())
    assert int(b1["concept_depth"].max()) < 32
    assert b1["token_in"].shape == b1["token_out"].shape and bool((b1["token_in"][0] == synth.TOK_STR).all())
    assert b1["cp_seq"].shape == (n - 1, B)
    # distinct ranks draw distinct graphs
    b3, _ = synth.make_config_batch("C1", rank=1)
    assert not torch.equal(b1["concept"], b3["concept"])


def test_synth_padded_and_eval_batches():
    from gtos_amd import synth
    b, s = synth.make_batch(77, 5, 12, 9, padded=True)
    lens = (b["concept"] != 0).sum(0)
    assert int(lens.min()) < int(lens.max()) == s["n"]
    for g in range(5):
        L = int(lens[g])
        assert bool((b["relation"][L:, :, g] == 0).all()) and bool((b["relation"][:, L:, g] == 0).all())
    e, _ = synth.make_batch(77, 5, 12, 9, train=False, extra_frac=1.0)
    assert e["relation"].dim() == 4 and int(e["relation"][0, 0, 0, 0]) == 3
    assert bool((e["relation_bank"][:, 0] == 0).all())             # type 0 = <PAD> in eval batches


def test_splitmix_reference_values():
    from gtos_amd.synth import SplitMix64
    # first outputs of splitmix64 seeded with 1234567 (published reference sequence)
    r = SplitMix64(1234567)
    got = [int(v) for v in r.u64(3)]
    assert got == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_lr_schedule_and_decay_rule():
    from gtos_amd.flat import inverse_sqrt_lr, is_no_decay
    assert abs(inverse_sqrt_lr(512, 1, 2000) - 512 ** -0.5 * 2000 ** -1.5) < 1e-15
    assert abs(inverse_sqrt_lr(512, 2000, 2000) - 512 ** -0.5 * 2000 ** -0.5) < 1e-15
    assert inverse_sqrt_lr(512, 8000, 2000) < inverse_sqrt_lr(512, 2000, 2000)
    assert is_no_decay("graph_encoder.layers.0.fc1.bias") and is_no_decay("token_embed_layer_norm.weight")
    assert not is_no_decay("graph_encoder.layers.0.self_attn.in_proj_weight")


def test_attention_shape_boundary_is_refused_at_construction():
    """include/gtos_hip.h documents the kernels' shape boundary (d % H == 0 like the reference, head width a multiple of 8
    and at most 512); the modules must raise at construction for anything else."""
    import pytest
    from gtos_amd._lib import GtosHipError
    from gtos_amd.graph_transformer import RelationMultiheadAttention
    from gtos_amd.transformer import MultiheadAttention
    for cls in (RelationMultiheadAttention, MultiheadAttention):
        for d, H in ((512, 8), (64, 1), (768, 8), (1024, 8), (96, 4), (48, 6), (640, 5)):
            cls(d, H)
        for d, H in ((32, 8), (36, 3), (1040, 1)):
            with pytest.raises(GtosHipError):
                cls(d, H)
        with pytest.raises(AssertionError):
            cls(100, 8)                         # the reference's own assert: embed_dim divisible by num_heads


def test_reference_module_names_resolve_to_the_product_modules():
    """The reference's bare import lines (generator/generator.py:6-9 and friends) resolve to gtos_amd after one call."""
    import sys
    import gtos_amd
    saved = {n: sys.modules.get(n) for n in gtos_amd.REFERENCE_MODULE_NAMES}
    try:
        for n in gtos_amd.REFERENCE_MODULE_NAMES:
            sys.modules.pop(n, None)
        assert gtos_amd.install_reference_names() == list(gtos_amd.REFERENCE_MODULE_NAMES)
        from graph_transformer import GraphTransformer, GraphTransformerLayer, RelationMultiheadAttention   # noqa: F401
        from transformer import Transformer, TransformerLayer, MultiheadAttention, SinusoidalPositionalEmbedding, SelfAttentionMask  # noqa: F401
        from encoder import RelationEncoder, TokenEncoder                                                   # noqa: F401
        from decoder import DecodeLayer, TokenGenerator                                                     # noqa: F401
        from generator import Generator
        from search import Beam                                                                              # noqa: F401
        import gtos_amd.generator as g
        assert Generator is g.Generator
        import types
        sys.modules["encoder"] = types.ModuleType("encoder")          # a foreign module of that name is left alone ...
        assert "encoder" not in gtos_amd.install_reference_names()
        assert "encoder" in gtos_amd.install_reference_names(force=True)   # ... unless forced
    finally:
        gtos_amd.uninstall_reference_names()
        for n, m in saved.items():
            if m is not None:
                sys.modules[n] = m


def test_auxiliary_stream_policy_follows_the_peak_allocation(monkeypatch):
    """ops.SIDE_STREAMS="auto": the auxiliary stream is used while a step's sampled peak allocation (ops.note_memory at the step's
    high-water points) stays below a quarter of the device memory; the decision is taken once per step from host-side allocator
    counters, FROM THE PREVIOUS STEP (not from torch's process-lifetime peak, which one large evaluation batch would poison and a user
    may reset), and it comes back with hysteresis once the steps are small again (C5: 93 GB allocated -> one allocator pool, 120 GB
    reserved instead of 230-277)."""
    import types

    import torch

    from gtos_amd import ops
    dev = torch.device("cuda", 0)
    cur = [10 << 30]
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: types.SimpleNamespace(total_memory=288 << 30))
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda d: cur[0])
    monkeypatch.setattr(torch.cuda, "max_memory_allocated", lambda d: 250 << 30)      # (nobody reads the process-wide peak any more)
    monkeypatch.setattr(ops, "_SIDE_POLICY", {})
    monkeypatch.setattr(ops, "_DEVICE_BYTES", {})
    monkeypatch.setattr(ops, "_STEP_PEAK", {})
    monkeypatch.setattr(ops, "SIDE_STREAMS", "auto")
    assert ops.side_ok(dev)                                  # before the first step: yes
    ops.refresh_side_policy(dev)
    assert ops.side_ok(dev)
    cur[0] = 93 << 30
    ops.note_memory(dev)                                     # a high-water point inside the step ...
    cur[0] = 5 << 30
    ops.note_memory(dev)                                     # ... a later, lower sample does not lower it
    assert ops.side_ok(dev)                                  # nothing changes inside a step
    ops.refresh_side_policy(dev)
    assert not ops.side_ok(dev)                              # the NEXT step runs without the stream
    cur[0] = 65 << 30                                        # below the bound (72 GB) but inside the hysteresis band (> 57.6 GB): stays off
    ops.note_memory(dev)
    ops.refresh_side_policy(dev)
    assert not ops.side_ok(dev)
    cur[0] = 40 << 30                                        # small steps again: the stream comes back
    ops.note_memory(dev)
    ops.refresh_side_policy(dev)
    assert ops.side_ok(dev)
    ops.refresh_side_policy(torch.device("cpu"))             # no-op
    for forced, want in (("0", False), ("1", True)):
        monkeypatch.setattr(ops, "_SIDE_POLICY", {})
        monkeypatch.setattr(ops, "SIDE_STREAMS", forced)
        assert ops.side_ok(dev) is want
        cur[0] = 200 << 30
        ops.note_memory(dev)
        ops.refresh_side_policy(dev)
        assert ops.side_ok(dev) is want
