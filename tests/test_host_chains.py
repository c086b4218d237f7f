"""The hybrid form's host-core stream coders (draco-oxide_amd/csrc/host_chains.cpp) against the oracle's restated
RansCoder / RabsCoder (encode/entropy/rans.rs:33-68, :91-128), byte for byte.  CPU only: dmi_host_rans_stream and
dmi_host_rabs_stream involve no device."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc


def _normalised(rng, n_sym, precision, zeros=0.0, tail=1.0):
    """Random normalised frequencies summing to 2^precision (every listed symbol ≥ 1 unless zeroed)."""
    w = rng.pareto(tail, size=n_sym) + 1e-3
    dead = rng.random(n_sym) < zeros
    dead[rng.integers(0, n_sym)] = False
    w[dead] = 0
    f = np.floor(w / w.sum() * (1 << precision)).astype(np.int64)
    f[(~dead) & (f == 0)] = 1
    diff = (1 << precision) - int(f.sum())
    k = int(np.argmax(f))
    if f[k] + diff < 1:
        pytest.skip("degenerate draw")
    f[k] += diff
    return f.astype(np.uint32)


@pytest.mark.parametrize("precision,n_sym,n,seed", [(12, 3, 1000, 1), (12, 200, 20000, 2), (13, 400, 30000, 3), (15, 900, 50000, 4), (16, 2000, 60000, 5),
                                                    (18, 3000, 80000, 6), (19, 6000, 80000, 7), (20, 1425, 200000, 8), (20, 16000, 200000, 9), (12, 1, 50, 10)])
def test_host_rans_stream_equals_the_oracle_coder(precision, n_sym, n, seed):
    rng = np.random.default_rng(seed)
    f = _normalised(rng, n_sym, precision, zeros=0.2 if n_sym > 8 else 0.0)
    p = f.astype(np.float64) / f.sum()
    syms = rng.choice(n_sym, size=n, p=p).astype(np.uint32)
    # rare symbols (f < 2^(P-8), several renormalisation bytes) and frequency-1 symbols in force
    live = np.nonzero(f)[0]
    rare = live[np.argsort(f[live])[: max(1, len(live) // 10)]]
    syms[rng.integers(0, n, size=max(1, n // 50))] = rng.choice(rare, size=max(1, n // 50))
    got = dmi.host_rans_stream(f, precision, syms)
    want = orc.rans_encode_raw(f.astype(np.uint64), precision, np.ascontiguousarray(syms[::-1]))   # the reference feeds the symbols in reverse (symbol_coding.rs:161-163)
    assert got == want
    back = orc.rans_decode_raw(got, f.astype(np.uint64), precision, n)
    assert np.array_equal(back, syms)


@pytest.mark.parametrize("zero_prob", [1, 2, 17, 128, 200, 254, 255])
@pytest.mark.parametrize("n", [0, 1, 63, 5000])
def test_host_rabs_stream_equals_the_oracle_coder(zero_prob, n):
    rng = np.random.default_rng(zero_prob * 131 + n)
    bits = (rng.random(n) >= zero_prob / 256.0).astype(np.uint8)
    if n > 10:
        bits[rng.integers(0, n, size=n // 7)] ^= 1
    got = dmi.host_rabs_stream(zero_prob, bits)
    assert got == orc.rabs_encode(zero_prob, bits)


def test_host_rans_stream_rejects_what_the_reference_cannot_code():
    f = np.array([4096 - 5, 0, 5], np.uint32)
    with pytest.raises(dmi.DracoMiError):
        dmi.host_rans_stream(f, 12, np.array([0, 1, 2], np.uint32))       # symbol 1 has no frequency
    with pytest.raises(dmi.DracoMiError):
        dmi.host_rans_stream(np.array([10, 20], np.uint32), 12, np.array([0], np.uint32))   # does not sum to 2^12


@pytest.mark.parametrize("zero_prob", [255, 254, 250, 200, 129, 128, 64, 3, 2, 1])
@pytest.mark.parametrize("bit", [0, 1])
def test_constant_bit_stream_by_its_period_equals_the_stepping_coder(zero_prob, bit):
    """dmi_host_rabs_constant_stream (the all-zero seam-flag stream of the connectivity stage): n copies of one bit coded as prefix + repeated
    period + tail — byte for byte what the stepping coder (pinned against the oracle's above) writes, across the window edge, whole periods,
    period ± 1, and streams too short to repeat."""
    lengths = [0, 1, 2, 100, 1409, 1410, 1411, 2819, 16383, 16384, 16385, 16384 + 1409, 16384 + 1410, 16384 + 3 * 1409 + 7, 100000, 1500007]
    for n in lengths:
        bits = np.full(n, bit, np.uint8)
        try:
            want = dmi.host_rabs_stream(zero_prob, bits)
        except dmi.DracoMiError:
            with pytest.raises(dmi.DracoMiError):
                dmi.host_rabs_constant_stream(zero_prob, bit, n)
            continue
        assert dmi.host_rabs_constant_stream(zero_prob, bit, n) == want, f"zero_prob {zero_prob}, bit {bit}, n {n}"
