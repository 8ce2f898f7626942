"""CPU: the oracle is pinned to the reference's own text.

``oracle/ref_extract.py`` executes the fenced pyspec code of /root/reference/pos-evolution.md inside ``oracle.spec``;
these tests hold the three links of that chain:

1. the Markdown's pinned fences are the ones whose sha256 is committed in tests/golden/ref_pins.json (one changed
   character in a pinned function fails here) -- against /root/reference when it is on this machine, and always
   against the generated oracle/_ref/MANIFEST.json;
2. what ``oracle.spec`` exports for every pinned name IS the code compiled from those fences;
3. ``oracle/spec.py``'s own transcription of each pinned function equals the reference text AST for AST, modulo the
   integer-cast stand-ins listed in ``ref_extract._Normalise`` -- so the fallback cannot drift either;
and that both oracles reproduce the committed fixtures.
"""
import ast
import dataclasses
import hashlib
import json
import os
import subprocess
import sys

import pytest

from oracle import ref_extract, spec

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PINS = json.load(open(os.path.join(HERE, "golden", "ref_pins.json")))


def test_pins_cover_every_hot_path_function_the_reference_defines():
    assert sorted(PINS["pinned"]) == sorted(ref_extract.PINNED_NAMES)
    # the line ranges SURVEY.md 8(a) / VERDICT r1 cite
    want = {"get_head": [1102, 1116], "update_latest_messages": [1435, 1441], "on_attestation": [963, 979],
            "process_attestation": [721, 755], "compute_shuffled_index": [512, 535], "compute_committee": [494, 506],
            "on_tick": [934, 955], "on_block": [986, 1036], "weigh_justification_and_finalization": [817, 852]}
    for name, (lo, hi) in want.items():
        a, b = PINS["pinned"][name]["pe"]
        assert a <= lo + 1 and b >= hi - 1, (name, a, b)


@pytest.mark.skipif(not ref_extract.reference_available(), reason="/root/reference is not on this machine")
def test_reference_markdown_matches_committed_pins():
    """Fails if one character of a pinned fence (or the fence inventory) of the Markdown changes."""
    assert ref_extract.pins_from_reference() == PINS


def test_generated_oracle_matches_committed_pins():
    assert ref_extract.available() or ref_extract.generate(), \
        "oracle/_ref/ missing: run __graft_entry__.build() where /root/reference exists"
    man = json.load(open(ref_extract.MANIFEST))
    assert man["markdown_sha256"] == PINS["markdown_sha256"]
    for name in ref_extract.PINNED_NAMES:
        assert man["pinned"][name]["sha256"] == PINS["pinned"][name]["sha256"], name
    # and the generated file holds exactly those texts (except the stitched on_attestation)
    text = open(ref_extract.FENCES_PY, encoding="utf-8").read()
    blocks = text.split("\n# ==== ")[1:]
    assert len(blocks) == len(ref_extract.PINNED_NAMES)
    for blk, name in zip(blocks, ref_extract.PINNED_NAMES):
        header, code = blk.split(" ====\n", 1)
        assert header.split()[0] == name
        if not man["pinned"][name]["derived"]:
            assert hashlib.sha256(code.encode()).hexdigest() == PINS["pinned"][name]["sha256"], name


def test_oracle_of_record_is_the_reference_text():
    assert spec.ORACLE_OF_RECORD == "reference"
    for name, kind in ref_extract.PINNED:
        obj = getattr(spec, name)
        if name == "update_latest_messages":
            obj = obj.literal  # wrapped by the vote-expiry variant's slot recorder (a straight call when eta == 0)
        if kind == "F":
            assert obj.__code__.co_filename == ref_extract.FENCES_PY, name
            assert obj.__globals__ is vars(spec), name  # undefined callees resolve to spec's [UPSTREAM-MEMORY] ones
        else:
            assert dataclasses.is_dataclass(obj), name
    assert [f.name for f in dataclasses.fields(spec.LatestMessage)] == ["epoch", "root"]
    assert [f.name for f in dataclasses.fields(spec.Store)] == [
        "time", "genesis_time", "justified_checkpoint", "finalized_checkpoint", "best_justified_checkpoint",
        "proposer_boost_root", "equivocating_indices", "blocks", "block_states", "checkpoint_states", "latest_messages"]


def test_transcription_equals_reference_ast():
    """oracle/spec.py's [REF] functions == the reference's, AST for AST, after removing docstrings/annotations and the
    integer-cast stand-ins (uint64(x) -> x, uint_to_bytes(uint8(x)) -> uint_to_bytes(x, 1), hash -> sha256)."""
    spec_src = open(os.path.join(ROOT, "oracle", "spec.py"), encoding="utf-8").read()
    ref_src = open(ref_extract.FENCES_PY, encoding="utf-8").read()
    for name in ref_extract.PINNED_NAMES:
        mine = ref_extract.normalised_dump(ref_extract.find_def(spec_src, name))
        theirs = ref_extract.normalised_dump(ref_extract.find_def(ref_src, name))
        assert mine == theirs, f"{name}: transcription differs from the reference text\n" \
            f"{ast.unparse(ref_extract._Normalise().visit(ref_extract.find_def(spec_src, name)))}\n--- vs ---\n" \
            f"{ast.unparse(ref_extract._Normalise().visit(ref_extract.find_def(ref_src, name)))}"


def test_normaliser_is_not_vacuous():
    """A one-token change of a pinned function must break the AST equality."""
    ref_src = open(ref_extract.FENCES_PY, encoding="utf-8").read()
    good = ref_extract.normalised_dump(ref_extract.find_def(ref_src, "update_latest_messages"))
    bad_src = ref_src.replace("target.epoch > store.latest_messages[i].epoch", "target.epoch >= store.latest_messages[i].epoch")
    assert bad_src != ref_src
    assert ref_extract.normalised_dump(ref_extract.find_def(bad_src, "update_latest_messages")) != good
    bad_src = ref_src.replace("uint_to_bytes(uint8(current_round))", "uint_to_bytes(uint32(current_round))")
    assert ref_extract.normalised_dump(ref_extract.find_def(bad_src, "compute_shuffled_index")) != \
        ref_extract.normalised_dump(ref_extract.find_def(ref_src, "compute_shuffled_index"))


def test_container_fields_follow_the_reference():
    """SSZ containers are not executed (the SSZ machinery is a stand-in) but spec.py's dataclasses carry the
    reference's fields in the reference's order; extra trailing fields are the documented stand-ins."""
    man = json.load(open(ref_extract.MANIFEST))["containers"]
    extra = {"Attestation": ["signature_valid"], "Validator": [], "Checkpoint": [], "AttestationData": [],
             "BeaconBlock": [], "AttesterSlashing": []}
    for name, info in man.items():
        assert info["sha256"] == PINS["containers"][name]["sha256"]
        mine = [f.name for f in dataclasses.fields(getattr(spec, name))]
        assert mine == info["fields"] + extra[name], (name, mine, info["fields"])


def test_reference_run_reproduces_committed_fixtures_and_so_does_the_transcription():
    """tests/golden/{forkchoice_trace,shuffle_vectors}.json were written by the reference's own code; regenerating
    them here (reference mode) and in a subprocess with the transcription forced must give the same bytes."""
    gen = os.path.join(HERE, "golden", "generate.py")
    committed = hashlib.sha256(json.dumps(
        {"trace": json.load(open(os.path.join(HERE, "golden", "forkchoice_trace.json"))),
         "shuffle": json.load(open(os.path.join(HERE, "golden", "shuffle_vectors.json")))}, sort_keys=True).encode()).hexdigest()
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.check_output([sys.executable, gen, "--digest"], env=env, text=True).split()
    assert out == ["reference", committed]
    env["POSEVO_ORACLE_TRANSCRIPTION_ONLY"] = "1"
    out = subprocess.check_output([sys.executable, gen, "--digest"], env=env, text=True).split()
    assert out == ["transcription", committed]


def test_typed_uint_to_bytes_widths():
    """The pyspec's uint_to_bytes takes the width from the SSZ type: pe:522 hashes 1 byte of the round, pe:525 4 bytes
    of position // 256, pe:486 8 bytes of the epoch.  The stand-ins must honour that for the reference text to run."""
    assert spec.uint_to_bytes(spec.uint8(7)) == b"\x07"
    assert spec.uint_to_bytes(spec.uint32(7)) == b"\x07\x00\x00\x00"
    assert spec.uint_to_bytes(spec.uint64(7)) == (7).to_bytes(8, "little") == spec.uint_to_bytes(7)
    assert spec.hash(b"abc") == hashlib.sha256(b"abc").digest()
    assert spec.hash((1, 2)) == hash((1, 2))


def test_generated_file_is_verified_against_committed_pins_before_exec():
    """oracle/_ref/ is git-ignored and regenerated from a file outside the repository; overlay() executes it
    only if every section hashes to its committed pin."""
    from oracle import ref_extract as r

    pins = r.committed_pins()
    assert pins is not None and "stitched_sha256" in pins["pinned"]["on_attestation"]
    if not r.available():
        pytest.skip("oracle/_ref/ not generated here")
    text = open(r.FENCES_PY, encoding="utf-8").read()
    assert r.verify_generated(text, pins) is None
    assert r.verify_generated(text.replace("store.time", "store.time ", 1), pins) is not None      # one byte changed
    assert r.verify_generated("import os\n" + text, pins) == "code before the first pinned section"
    assert r.verify_generated(text + "\n# ==== extra  pe:1-2 ====\nprint(1)\n", pins) is not None   # unpinned section
    bad = dict(pins, markdown_sha256="0" * 64)
    assert r.verify_generated(text, bad) is None  # sections are what is executed; the markdown hash gates generate()
