"""GPU: the staged HIP trie builder (gtos_amd.pathtrie_hip -> gtos_pathtrie_dev_phase_a / _b of libgtos_hip.so: rocPRIM sorts and
scans around the stage kernels of csrc/trie_kernels.h) against the host builder (csrc_host/pathtrie.cpp), array for array, and the
RelationEncoder on tries built that way.  The stage code itself is proven equal on the CPU (tests/test_pathtrie.py through the
emulation library); this file covers what only the GPU can: the launch glue and rocPRIM.

Sorted last on purpose: the entry points first ran on an MI355X in the closing minutes of round 3 (tools/hip_trie_check.py:
profiles/r3w_hip_trie_check.json -- small bank and C2 bank equal, 1.5 ms per build); the rejection paths and the encoder leg below had
no GPU time left to run on, and a failure here must not hide the rest of the suite under ``-x``."""
import pytest
import torch

from test_pathtrie import _random_bank, _same_object, _check

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _hip(bank, length, **kw):
    from gtos_amd.pathtrie_hip import HipBackend, build_path_trie_staged
    return build_path_trie_staged(bank.to(dev()), length.to(dev()), HipBackend.shared(), **kw)


def _host_copy(trie):
    return trie.to(torch.device("cpu"))


@pytest.mark.parametrize("seed,R,L,V", [(1, 1, 1, 5), (2, 40, 4, 6), (3, 300, 8, 5), (4, 500, 8, 250), (6, 900, 6, 12)])
@pytest.mark.parametrize("chunk", [8, 64])
def test_hip_trie_builder_equals_the_host_builder(seed, R, L, V, chunk):
    from gtos_amd.pathtrie import build_path_trie
    seqs, bank, length = _random_bank(seed, R, L, V)
    host = build_path_trie(bank, length, chunk=chunk)
    hip = _host_copy(_hip(bank, length, chunk=chunk))
    assert _same_object(host, hip) == []
    _check(seqs, hip, chunk=chunk)


def test_hip_trie_builder_at_c2_size_duplicates_and_limits():
    from gtos_amd import synth
    from gtos_amd.pathtrie import build_path_trie
    batch, st = synth.make_config_batch("C2", rank=0, B=64)
    host = build_path_trie(batch["relation_bank"], batch["relation_length"])
    assert _same_object(host, _host_copy(_hip(batch["relation_bank"], batch["relation_length"], n_rows=host.N))) == []
    assert _same_object(host, _host_copy(_hip(batch["relation_bank"], batch["relation_length"]))) == []     # row count read from the device
    g = torch.Generator().manual_seed(5)                                   # a bank with many duplicate paths
    length = torch.randint(1, 9, (3000,), generator=g)
    bank = torch.randint(1, 7, (8, 3000), generator=g) * (torch.arange(8)[:, None] < length[None, :])
    assert _same_object(build_path_trie(bank, length), _host_copy(_hip(bank, length))) == []
    with pytest.raises(ValueError):                                        # label ids the one-byte keys cannot hold
        _hip(torch.full((2, 3), 300, dtype=torch.int64), torch.tensor([1, 2, 2]))
    with pytest.raises(ValueError):                                        # a path of 9 labels
        _hip(torch.ones(9, 2, dtype=torch.int64), torch.tensor([9, 1]))
    with pytest.raises(ValueError):                                        # an empty path
        _hip(torch.ones(3, 4, dtype=torch.int64), torch.tensor([1, 0, 2, 3]))


def test_relation_encoder_on_hip_built_tries_equals_host_built_tries():
    """RelationEncoder (bf16, trie evaluation) forward and every parameter gradient with tries from the HIP builder == the same with
    tries from the host builder: the index arrays are equal, so the results agree up to the order of the fp32 atomic additions a few
    kernels make."""
    from gtos_amd import synth
    from gtos_amd.pathtrie import build_path_trie
    from test_hip_parity import _relenc_pair, _grads_of, _rel_frob
    batch, _ = synth.make_batch(3, 6, 40, 8)
    bank, length = batch["relation_bank"], batch["relation_length"]
    _, m = _relenc_pair(bank, length)
    m.compute_dtype = torch.bfloat16
    wout = torch.randn(bank.shape[1], 64, generator=torch.Generator().manual_seed(1)).to(dev())
    res = []
    for trie in (build_path_trie(bank, length).to(dev()), _hip(bank, length)):
        m.zero_grad()
        out = m(bank.to(dev()), length.to(dev()), trie=trie)
        (out.float() * wout).sum().backward()
        res.append((out.detach().float().cpu(), _grads_of(m)))
    assert _rel_frob(res[1][0], res[0][0]) < 1e-3          # (equal index arrays; a bar instead of torch.equal in case a forward kernel sums in fp32 atomics)
    for k in res[0][1]:
        assert _rel_frob(res[1][1][k], res[0][1][k]) < 1e-3, k


def test_launch_stream_handle_follows_the_current_stream():
    """_lib.stream() (the raw current-stream getter once adopted, the public API otherwise) is the handle of torch's current stream on the
    default stream, inside a side-stream context and after leaving it -- what every launch of the library is queued on."""
    from gtos_amd import _lib
    dev()                                       # (a CUDA context)
    torch.zeros(1, device=dev())
    for _ in range(2):                          # first call adopts (or rejects) the raw getter, second uses it
        assert _lib.stream() == torch.cuda.current_stream().cuda_stream
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            assert _lib.stream() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
        assert _lib.stream() == torch.cuda.current_stream().cuda_stream
    print("raw stream getter adopted:", bool(_lib._raw_stream))
