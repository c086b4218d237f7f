"""Pins the CPU restatement (oracle/) against every known-answer test the reference holds for the
attribute-encoding path.  Each test names the reference test it replays."""
import os

import numpy as np
import pytest

import orc

DATA = os.path.join(os.path.dirname(__file__), "golden", "data")


def obj(name, faithful=True):
    return orc.Session.from_obj(os.path.join(DATA, name + ".obj"), faithful=faithful)


def test_leb128_manual():
    # utils/bit_coder.rs:41-49  manual_test_leb128_write_read
    assert orc.leb128(300) == bytes([172, 2])
    for v in [0, 1, 127, 128, 255, 256, 1234567890, 0xFFFFFFFFFFFFFFFF]:   # :52-66 more_tests_leb128
        b = orc.leb128(v)
        r, sh = 0, 0
        for x in b:
            r |= (x & 0x7F) << sh
            sh += 7
        assert r == v and (b[-1] & 0x80) == 0


def test_bitwriter_msb_first():
    # core/bit_coder.rs:511-592 test_writer_reader_msb_first
    assert len(orc.bitwriter([(2, 0b10), (3, 0b011)], msb=True)) == 1
    assert len(orc.bitwriter([(7, 0b0111010)], msb=True)) == 1
    assert orc.bitwriter([(8, 0b10111010)], msb=True) == bytes([0b10111010])
    assert orc.bitwriter([(9, 0b110111011)], msb=True) == bytes([0b11011101, 0b10000000])
    b = orc.bitwriter([(9, 0b101010100), (8, 0b10101110), (7, 0b0101010), (6, 0b111100), (5, 0b00001), (4, 0b1100)], msb=True)
    assert len(b) == (9 + 8 + 7 + 6 + 5 + 4) // 8 + 1
    assert list(b[:5]) == [0b10101010, 0b01010111, 0b00101010, 0b11110000, 0b00111000]
    assert len(orc.bitwriter([(11, 0b10111010110)], msb=True)) == 2


def _read_lsb(buf, sizes):
    bits = []
    for byte in buf:
        for k in range(8):
            bits.append((byte >> k) & 1)
    out, p = [], 0
    for s in sizes:
        v = 0
        for k in range(s):
            v |= bits[p + k] << k
        out.append(v)
        p += s
    return out


def test_bitwriter_lsb_first():
    # core/bit_coder.rs:594-627 test_writer_reader_lsb_first
    vals = [(9, 0b101010100), (8, 0b10101010), (7, 0b0101010), (6, 0b111100), (5, 0b00001), (4, 0b1100)]
    b = orc.bitwriter(vals, msb=False)
    assert len(b) == (9 + 8 + 7 + 6 + 5 + 4) // 8 + 1
    assert _read_lsb(b, [s for s, _ in vals]) == [v for _, v in vals]
    b = orc.bitwriter([(10, 0b1010101010)], msb=False)
    assert len(b) == 2 and _read_lsb(b, [2] * 5) == [0b10] * 5


def test_obj_tetrahedron_indexing():
    # io/obj/mod.rs:73-88 tetrahedron
    s = obj("tetrahedron")
    assert s.faces().tolist() == [[0, 1, 2], [0, 3, 1], [0, 2, 4], [1, 5, 2]]
    atts = s.attributes()
    assert len(atts) == 3
    assert atts[0]["type"] == orc.POSITION and atts[0]["domain"] == orc.DOM_POSITION
    assert atts[0]["ncomp"] == 3 and atts[0]["num_unique"] == 4 and atts[0]["len"] == 6


def test_attribute_remap():
    # core/attribute/mod.rs:788-815 test_attribute_remap: dedup map [0,1,2,0,1,3]
    pos = np.array([[0, 0, 0], [1, 0, 0], [0.5, 1, 0], [0, 0, 0], [1, 0, 0], [2, 0, 0]], np.float32)
    for faithful in (True, False):
        s = orc.Session()
        s.L.orc_builder_reset()
        par = np.zeros(0, np.uint32)
        s.L.orc_builder_add_attribute(orc._ptr(pos), 6, orc.POSITION, 0, orc.F32, 3, orc._ptr(par), 0, int(faithful))
        # a face list that references every point and produces no point-level duplicates is not
        # needed here: inspect the pending attribute through a trivial build with distinct points
        f = np.array([[0, 1, 2], [3, 4, 5]], np.uint32)
        s.L.orc_builder_set_faces(orc._ptr(f), 2)
        s._check(s.L.orc_build(s.h, int(faithful)))
        # points 3,4 are byte-identical to 0,1 → MeshBuilder merges them (builder.rs:194-250)
        a = s.attributes()[0]
        assert a["num_unique"] == 4
        assert a["p2v"] is not None and a["p2v"].tolist() == [0, 1, 2, 3]
        assert s.faces().tolist() == [[0, 1, 2], [0, 1, 3]]


def test_value_dedup_modes_agree():
    rng = np.random.default_rng(1)
    base = rng.integers(0, 8, size=(200, 3)).astype(np.float32)
    base[5] = [0.0, -0.0, 1.0]
    base[9] = [-0.0, 0.0, 1.0]
    faces = rng.integers(0, 200, size=(300, 3)).astype(np.uint32)
    out = []
    for faithful in (True, False):
        s = orc.Session.from_arrays(faces, [dict(data=base, type=orc.POSITION)], faithful=faithful)
        a = s.attributes()[0]
        out.append((s.faces().tobytes(), a["data"].tobytes(), None if a["p2v"] is None else a["p2v"].tobytes()))
    assert out[0] == out[1]


def test_mesh_builder_tetrahedron():
    # core/mesh/builder.rs:406-436 test_with_tetrahedron: 12 points dedup to 4
    faces = np.arange(12, dtype=np.uint32).reshape(4, 3)
    x = [0, 1, 2, 0, 3, 1, 1, 3, 2, 0, 2, 3]
    pos = np.array([[v, 0, 0] for v in x], np.float32)
    s = orc.Session.from_arrays(faces, [dict(data=pos, type=orc.POSITION)], faithful=True)
    assert len(s.faces()) == 4
    a = s.attributes()
    assert len(a) == 1 and a[0]["len"] == 4


def test_corner_table_two_triangles():
    # core/corner_table/mod.rs:539-583 test_corner_table
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    s = orc.Session.from_arrays([[0, 1, 2], [2, 1, 3]], [dict(data=pos, type=orc.POSITION)], faithful=True)
    s.encode(faithful=True)
    opp = s.blob("ct.opp", np.uint32)
    N = 0xFFFFFFFF
    assert opp.tolist() == [5, N, N, N, N, 0]
    assert s.blob("ct.nverts", np.uint32)[0] == 4
    assert s.blob("ct.c2v", np.uint32).tolist() == [0, 1, 2, 2, 1, 3]


def test_corner_table_triangle():
    # core/corner_table/mod.rs:614-632 test_triangle
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    s = orc.Session.from_arrays([[0, 1, 2]], [dict(data=pos, type=orc.POSITION)], faithful=True)
    s.encode(faithful=True)
    assert s.blob("ct.lmc", np.uint32).tolist() == [0, 1, 2]
    assert s.blob("ct.nverts", np.uint32)[0] == 3


def test_corner_table_non_manifold_vertex():
    # core/corner_table/mod.rs:634-659 test_non_manifold: vertex 0 is split, 6 vertices, lmc [0,1,2,4,5,3]
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [-1, 1, 0], [0, -1, 0]], np.float32)
    s = orc.Session.from_arrays([[0, 1, 2], [0, 3, 4]], [dict(data=pos, type=orc.POSITION)], faithful=True)
    s.encode(faithful=True)
    assert s.blob("ct.nverts", np.uint32)[0] == 6
    assert s.blob("ct.lmc", np.uint32).tolist() == [0, 1, 2, 4, 5, 3]


def test_no_att_seam_sphere():
    # core/corner_table/attribute_corner_table.rs:200-241 test_no_att_seam
    s = obj("sphere")
    s.encode(faithful=True)
    assert not s.blob("at0.seam").any()
    assert (s.blob("at0.opp", np.uint32) == s.blob("ct.opp", np.uint32)).all()
    assert (s.blob("at0.c2v", np.uint32) == s.blob("ct.c2v", np.uint32)).all()
    assert len(s.blob("at0.lmc", np.uint32)) == s.blob("ct.nverts", np.uint32)[0]


def test_att_seam_tetrahedron():
    # core/corner_table/attribute_corner_table.rs:244-291 test_att_seam (UV table = att_tables[1])
    s = obj("tetrahedron")
    s.encode(faithful=True)
    nverts = int(s.blob("ct.nverts", np.uint32)[0])
    lmc = s.blob("at1.lmc", np.uint32)
    assert len(lmc) == nverts + 2
    assert s.blob("at1.c2v", np.uint32)[0] == 0
    seam = s.blob("at1.seam")
    assert all(seam[c] for c in [3, 5, 6, 7, 9, 11])
    assert lmc.tolist() == [6, 5, 11, 10, 8, 4]
    opp = s.blob("at1.opp", np.uint32)
    nxt = lambda c: c - 2 if c % 3 == 2 else c + 1
    prv = lambda c: c + 2 if c % 3 == 0 else c - 1
    N = 0xFFFFFFFF
    for c in [4, 8, 10]:
        assert opp[nxt(c)] == N and opp[prv(c)] == N   # swing_left/right are None
    for c in lmc:
        assert opp[nxt(int(c))] == N


def test_traverser_tetrahedron():
    # shared/attribute/sequence.rs:163-207 test_traverser
    for faithful in (True, False):
        s = obj("tetrahedron", faithful)
        s.encode(faithful=faithful)
        c2p = s.blob("ct.c2p", np.uint32)
        assert c2p[s.blob("att0.seq", np.uint32)].tolist() == [3, 1, 0, 2]
        assert c2p[s.blob("att1.seq", np.uint32)].tolist() == [3, 1, 0, 2]
        assert c2p[s.blob("att2.seq", np.uint32)].tolist() == [3, 1, 0, 2, 5, 4]


def test_rans_roundtrip_reference_sequence():
    # decode/entropy/rans.rs:219-244 test_rans_decoder: x=(x+37)%43 ×4096, default precision 12
    num_symbols = 43
    data, freq = [], [0] * num_symbols
    x = 3
    for _ in range(1 << 12):
        x = (x + 37) % num_symbols
        data.append(x)
        freq[x] += 1
    enc = orc.rans_encode_raw(freq, 12, data)
    dec = orc.rans_decode_raw(enc, freq, 12, len(data))
    assert dec.tolist() == data[::-1]


def test_rabs_roundtrip_reference_sequence():
    # decode/entropy/rans.rs:246-280 test_rabs_coder
    n, num_zeros = 256, 100
    srt = [0] * num_zeros + [1] * (n - num_zeros)
    data = [0] * n
    for i in range(n):
        data[(67 * i) % n] = srt[i]
    enc = orc.rabs_encode(num_zeros, data)
    dec = orc.rabs_decode(enc, num_zeros, n)
    assert dec.tolist() == data[::-1]


@pytest.mark.parametrize("length", [100, 300])
def test_encode_decode_symbols_direct(length):
    # decode/entropy/symbol_coding.rs:163-210 test_encode_decode_symbols_direct_coded(_multi_components)
    syms = [(x * x * x) % 23 for x in range(length)]
    enc = orc.encode_symbols(syms)
    dec, used = orc.decode_symbols(enc, length)
    assert used == len(enc)
    assert dec.tolist() == syms


def test_octahedral_roundtrip_property():
    # encode/attribute/prediction_transform/geom.rs:167-195: the restated oct quantiser must invert to
    # the input direction within the 8-bit cell (the reference test uses the float transform; here the
    # quantised form is checked through the encoder's own output)
    vs = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 1, 1], [-1, -1, -1],
                   [1, -1, 1], [-1, 1, -1], [1, 1, -1], [-1, -1, 1], [1, -1, -1]], np.float32)
    vs = vs / np.linalg.norm(vs, axis=1, keepdims=True)
    pos = np.arange(13 * 3, dtype=np.float32).reshape(13, 3) ** 1.1
    faces = [[i, (i + 1) % 13, (i + 2) % 13] for i in range(0, 11)]
    s = orc.Session.from_arrays(faces, [dict(data=pos, type=orc.POSITION), dict(data=vs.astype(np.float32), type=orc.NORMAL, domain=orc.DOM_CORNER, parents=[0])])
    s.encode()
    q = s.blob("att1.q", np.int32).reshape(-1, 2)
    n = s.attributes()[1]["data"]
    for (a, b), v in zip(q, n):
        if (a, b) == (255, 255):
            continue
        u, w = a / 127.0 - 1.0, b / 127.0 - 1.0
        x = 1.0 - abs(u) - abs(w)
        y, z = u, w
        if abs(u) + abs(w) > 1.0:
            y = (1 - abs(w)) * (1 if u > 0 else -1)
            z = (1 - abs(u)) * (1 if w > 0 else -1)
        r = np.array([x, y, z])
        r /= np.linalg.norm(r)
        assert np.dot(r, v) > 0.99


@pytest.mark.parametrize("name", ["tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus"])
def test_faithful_and_ranked_modes_byte_identical(name):
    a = obj(name, True).encode(faithful=True)
    b = obj(name, False).encode(faithful=False)
    assert a == b
    assert a[:5] == b"DRACO" and a[5:11] == bytes([2, 2, 1, 1, 0, 0])   # encode/header/mod.rs:26-54
