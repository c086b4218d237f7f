"""The JSON layer of the native transcode driver (csrc/dmi_json.hpp behind dmi_json_roundtrip; host only): a document must survive parse + write —
members in document order, numbers as written — and damaged documents must be refused, not crash."""
import json
import os

import pytest

from draco_oxide_amd import binding, gltf, synth

DUCK = os.path.join(os.path.dirname(__file__), "golden", "data", "Duck.glb")


def _compact(doc):
    return json.dumps(doc, separators=(",", ":"), ensure_ascii=False).encode("utf-8")


def test_documents_written_by_the_interpreter_come_back_byte_for_byte():
    glb, _ = synth.torus_glb(12, seed=3)
    doc, _ = gltf.read_glb(glb)
    doc["extras"] = {"name": "tür \"quoted\" \\ back\nslash\ttab \x01 snow ☃ \U0001d11e", "empty": {}, "list": [], "nested": [[1, 2.5, -3e-07, 1e+20, True, False, None]],
                     "big": 18446744073709551615}
    text = _compact(doc)
    assert binding.json_roundtrip(text) == text


def test_escapes_and_white_space():
    src = b' { "a" : "\\u00e9\\ud834\\udd1e\\/\\b\\f" ,\n "b":[ 1 , 2 ]\t, "c" : -0.0 , "d": 1E5 } '
    out = binding.json_roundtrip(src)
    assert json.loads(out) == json.loads(src)
    assert out == '{"a":"é\U0001d11e/\\b\\f","b":[1,2],"c":-0.0,"d":1E5}'.encode("utf-8")   # number tokens as written


def test_duck_document_is_semantically_unchanged():
    data = open(DUCK, "rb").read()
    n = int.from_bytes(data[12:16], "little")
    text = data[20: 20 + n]
    assert json.loads(binding.json_roundtrip(text)) == json.loads(text)


@pytest.mark.parametrize("bad", [b"", b"{", b'{"a":}', b'{"a":1,}', b"[1 2]", b'{"a":"\\x"}', b'"\\ud800', b"01", b"1.", b"nul", b'{"a":1} x', b"[" * 300 + b"]" * 300, b'{"a":"\x01"}'])
def test_damaged_documents_are_refused(bad):
    with pytest.raises(binding.DracoMiError):
        binding.json_roundtrip(bad)
