"""parakeet_amd.frontend's English side against tests/golden/en_frontend.json: the reference's own normaliser / phonectic.py /
arpabet.py / vocab.py run over stand-ins for ``inflect`` and ``g2p_en`` (tools/make_golden_en.py; VERDICT r04 "next" #5a).
The engine-side number speller (frontend/normalizer.py) and the stand-in's were written independently: this is where they meet."""
import json
import os
import subprocess
import sys

import pytest

from en_cases import RECIPE_SENTENCES, SENTENCES
from parakeet_amd.frontend import normalizer as nz
from parakeet_amd.frontend.phonectic import ARPABET, ARPABETWithStress, English, EnglishCharacter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "golden", "en_frontend.json")
GOLD = json.load(open(PATH, encoding="utf-8"))


def test_fixture_covers_the_case_list():
    assert len(SENTENCES) >= 50 and len(set(SENTENCES)) == len(SENTENCES)
    for section in ("normalize_numbers", "normalize", "english", "character", "arpabet", "arpabet_stress"):
        assert sorted(GOLD[section]) == sorted(SENTENCES), section


def test_number_normaliser_matches_the_reference_source():
    for s in SENTENCES:
        assert nz.normalize_numbers(s) == GOLD["normalize_numbers"][s], s


def test_sentence_normaliser_matches_the_reference_source():
    for s in SENTENCES:
        assert nz.normalize(s) == GOLD["normalize"][s], s
    for s, (half, full) in GOLD["width"].items():
        assert nz.full2half_width(s) == half and nz.half2full_width(s) == full


def test_vocabularies_match_the_reference_source():
    en, ch, a0, a1 = English(), EnglishCharacter(), ARPABET(), ARPABETWithStress()
    v = GOLD["vocab"]
    assert list(en.vocab.stoi) == v["english"] and list(ch.vocab.stoi) == v["character"]
    assert list(a0.vocab.stoi) == v["arpabet"] and list(a1.vocab.stoi) == v["arpabet_stress"]
    assert [en.vocab_size, ch.vocab_size, a0.vocab_size, a1.vocab_size] == v["sizes"]
    assert [en.vocab.padding_index, en.vocab.unk_index, en.vocab.start_index, en.vocab.end_index] == v["special_indices"]


def test_english_phones_and_ids_match_the_reference_source():
    en = English()
    for s in SENTENCES:
        ref = GOLD["english"][s]
        phones = en.phoneticize(s)
        assert phones == ref["phones"], s
        assert en(s) == ref["ids"] and en.numericalize(phones) == ref["ids"] and en.reverse(ref["ids"]) == ref["phones"], s


def test_character_frontend_matches_the_reference_source():
    ch = EnglishCharacter()
    for s in SENTENCES:
        ref = GOLD["character"][s]
        assert ch.phoneticize(s) == ref["text"] and ch(s) == ref["ids"], s


@pytest.mark.parametrize("cls,key", [(ARPABET, "arpabet"), (ARPABETWithStress, "arpabet_stress")])
def test_arpabet_frontends_match_the_reference_source(cls, key):
    fe = cls()
    for s in SENTENCES:
        ref = GOLD[key][s]
        assert fe.phoneticize(s) == ref["phones"], s
        assert fe(s) == ref["ids"] and fe(s, add_start_end=True) == ref["with_start_end"], s


def test_recipe_id_mapping_matches_the_reference_loop():
    """examples/fastspeech2/ljspeech/synthesize_e2e.py:88-98 = parakeet_amd.frontend.text_to_ids (what examples/synthesize_e2e.py calls)."""
    from parakeet_amd.frontend import text_to_ids
    en = English()
    table = GOLD["recipe"]["phone_id_map"]
    phone_id_map = {p: i for i, p in enumerate(table)}
    for s in RECIPE_SENTENCES:
        ref = GOLD["recipe"]["sentences"][s]
        assert text_to_ids(en, s, phone_id_map).tolist() == ref["ids"], s


@pytest.mark.skipif(not os.path.isdir(os.path.join(os.environ.get("PARAKEET_REFERENCE", "/root/reference"), "parakeet")),
                    reason="reference checkout not present")
def test_fixture_regenerates_byte_identically(tmp_path):
    before = open(PATH, "rb").read()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_golden_en.py")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(PATH, "rb").read() == before
