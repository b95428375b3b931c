"""Host-side text frontend: vocabulary layout, normalisation, lexicon G2P, the recipe's id mapping.

The reference has no tests for its frontend; the facts pinned here are read off its source:
vocab.py:46-66 (specials first, in order), punctuation.py:18-28, phonectic.py:48-73 (vocabulary = g2p_en phonemes +
English punctuation, start / end symbols around the phones), normalizer.py:22-34, numbers.py:17-86 and
examples/fastspeech2/ljspeech/synthesize_e2e.py:88-97.
"""
import numpy as np
import pytest

from parakeet_amd.frontend import (English, EnglishCharacter, LexiconG2p, Vocab, get_punctuations, normalize,
                                   normalize_numbers, phones_to_ids, read_phone_id_map, text_to_ids,
                                   full2half_width, half2full_width)
from parakeet_amd.frontend.g2p import letter_to_sound
from parakeet_amd.frontend.normalizer import number_to_words


def test_vocab_layout_and_lookup():
    v = Vocab(["a", "b", "a", "c"])
    assert list(v.stoi) == ["<pad>", "<unk>", "<s>", "</s>", "a", "b", "c"]
    assert (v.padding_index, v.unk_index, v.start_index, v.end_index, v.num_specials) == (0, 1, 2, 3, 4)
    assert v.lookup("b") == 5 and v.reverse(6) == "c" and len(v) == 7
    with pytest.raises(KeyError):
        v.lookup("zz")
    v.add_symbols(["c", "d"])
    assert len(v) == 8 and v.lookup("d") == 7
    v2 = Vocab(["x"], start_symbol=None, end_symbol=None)      # falsy specials are skipped (vocab.py:47-50)
    assert list(v2.stoi) == ["<pad>", "<unk>", "x"] and v2.start_index == -1


def test_vocab_public_dicts_are_live_attributes():
    """the reference's ``stoi`` / ``itos`` / ``special_symbols`` are plain dicts its callers read and WRITE
    (parakeet/frontend/vocab.py:28-45,98-105): same objects on every access, writable, kept in step by add_symbol."""
    v = Vocab(["a", "b"])
    assert v.stoi is v.stoi and v.itos is v.itos and v.special_symbols is v.special_symbols
    assert v.itos == {0: "<pad>", 1: "<unk>", 2: "<s>", 3: "</s>", 4: "a", 5: "b"}
    assert v.special_symbols == {"<pad>": 0, "<unk>": 1, "<s>": 2, "</s>": 3} and v.num_specials == 4
    v.add_symbol("c")
    assert v.itos[6] == "c" and v.stoi["c"] == 6 and len(v) == 7
    v.stoi["zz"] = 7                      # reference-style mutation by a caller
    v.itos[7] = "zz"
    assert v.lookup("zz") == 7 and v.reverse(7) == "zz" and len(v) == 8 and "zz" in v
    with pytest.raises(KeyError):
        v.reverse(99)
    assert str(v) == repr(v) and repr(v).startswith("Vocab(size: 8,")


def test_punctuations():
    assert get_punctuations("en") == [" ", "-", "...", ",", ".", "?", "!"]
    assert get_punctuations("cn") == ["、", "，", "；", "：", "。", "？", "！"]
    with pytest.raises(ValueError):
        get_punctuations("fr")


@pytest.mark.parametrize("n,words", [
    (0, "zero"), (7, "seven"), (13, "thirteen"), (21, "twenty-one"), (40, "forty"), (100, "one hundred"),
    (101, "one hundred one"), (342, "three hundred forty-two"), (3000, "three thousand"),
    (12003, "twelve thousand, three"), (1234567, "one million, two hundred thirty-four thousand, five hundred sixty-seven"),
])
def test_cardinals_without_andword(n, words):
    assert number_to_words(n, andword="") == words


@pytest.mark.parametrize("text,words", [
    ("1st", "first"), ("2nd", "second"), ("3rd", "third"), ("5th", "fifth"), ("12th", "twelfth"),
    ("20th", "twentieth"), ("21st", "twenty-first"), ("100th", "one hundredth"), ("101st", "one hundred and first"),
])
def test_ordinals(text, words):
    assert number_to_words(text) == words


def test_normalize_numbers_branches():
    # numbers.py:56-74: 1000 < n < 3000 is read as a year
    assert normalize_numbers("1984") == "nineteen eighty-four"
    assert normalize_numbers("1905") == "nineteen oh five"
    assert normalize_numbers("2000") == "two thousand"
    assert normalize_numbers("2007") == "two thousand seven"
    assert normalize_numbers("1900") == "nineteen hundred"
    assert normalize_numbers("3000") == "three thousand"
    assert normalize_numbers("12,345") == "twelve thousand, three hundred forty-five"
    assert normalize_numbers("3.14") == "three point fourteen"
    assert normalize_numbers("$3.50") == "three dollars, fifty cents"
    assert normalize_numbers("$1") == "one dollar"
    assert normalize_numbers("$0.01") == "one cent"
    assert normalize_numbers("£20") == "twenty pounds"
    assert normalize_numbers("the 3rd") == "the third"


def test_normalize_sentence():
    assert normalize("Café costs $2, i.e. 2 dollars (e.g. today)!") == \
        "cafe costs two dollars, that is two dollars for example today!"
    assert normalize("Hello — WORLD; ok?") == "hello  world ok?"


def test_width_conversion_round_trip():
    s = "Hello, World 123!"
    assert full2half_width(half2full_width(s)) == s
    assert half2full_width("A ") == "Ａ　"


def test_english_vocabulary_matches_the_reference_layout():
    en = English()
    # 4 specials + 70 stressed ARPAbet symbols (g2p_en's table) + 7 punctuation marks (phonectic.py:49-52)
    assert en.vocab_size == 81
    assert list(en.vocab.stoi)[:5] == ["<pad>", "<unk>", "<s>", "</s>", "AA0"]
    assert list(en.vocab.stoi)[-7:] == [" ", "-", "...", ",", ".", "?", "!"]
    assert en.vocab.lookup("UW") + 1 == en.vocab.lookup("UW0")   # g2p_en lists the bare UW before UW0..2


def test_english_phoneticize_and_ids():
    en = English()
    phones = en.phoneticize("Hello, world!")
    assert phones == ["<s>", "HH", "AH0", "L", "OW1", " ", ",", " ", "W", "ER1", "L", "D", " ", "!", "</s>"]
    ids = en(" Hello world")
    assert en.reverse(ids) == ["<s>", "HH", "AH0", "L", "OW1", " ", "W", "ER1", "L", "D", "</s>"]
    assert en.backend.oov == []
    en.phoneticize("zorblat")
    assert en.backend.oov == ["zorblat"]


def test_lexicon_file_format(tmp_path):
    lex = tmp_path / "lex.txt"
    lex.write_text(";;; comment\nTOMATO  T AH0 M EY1 T OW2\nTOMATO(2)  T AH0 M AA1 T OW2\n")
    g = LexiconG2p(str(lex))
    assert g("tomato tomato.") == ["T", "AH0", "M", "EY1", "T", "OW2", " ", "T", "AH0", "M", "EY1", "T", "OW2", " ", "."]
    bad = tmp_path / "bad.txt"
    bad.write_text("WORD  W QQ1\n")
    with pytest.raises(ValueError):
        LexiconG2p(str(bad))


def test_letter_to_sound_rules():
    assert letter_to_sound("make") == ["M", "EY1", "K"]
    assert letter_to_sound("phone") == ["F", "OW1", "N"]
    assert letter_to_sound("night") == ["N", "AY1", "T"]
    assert letter_to_sound("city") == ["S", "IH1", "T", "IY0"]
    assert letter_to_sound("queen") == ["K", "W", "IY1", "N"]
    valid = set(LexiconG2p.phonemes)
    for w in ("strength", "rhythm", "xylophone", "a", "zzz", "don't"):
        assert all(p in valid for p in letter_to_sound(w))


def test_english_character_frontend():
    ec = EnglishCharacter()
    assert ec.vocab_size == 37                       # <pad> <unk> <s> </s> + 26 letters + 7 punctuation marks
    assert ec.phoneticize("Hi, 2 you!") == "hi, two you!"
    assert "".join(ec.reverse(ec("Hi, 2 you!"))) == "hi, two you!"


def test_recipe_id_mapping(tmp_path):
    # synthesize_e2e.py:88-97: start / end dropped, spaces dropped, unknown phones and punctuation -> "sp"
    pm = tmp_path / "phone_id_map.txt"
    pm.write_text("<pad> 0\n<unk> 1\nHH 2\nAH0 3\nL 4\nOW1 5\nsp 6\n, 7\n<eos> 8\n")
    table = read_phone_id_map(str(pm))
    assert table["sp"] == 6
    phones = ["<s>", "HH", "AH0", "L", "OW1", " ", ",", " ", "W", "ER1", "</s>"]
    ids = phones_to_ids(phones, table)
    assert ids.dtype == np.int64
    assert ids.tolist() == [2, 3, 4, 5, 6, 6, 6]     # "," is in the map but is punctuation -> sp; W, ER1 unknown -> sp
    assert text_to_ids(English(), "Hello,", table).tolist() == [2, 3, 4, 5, 6]
    with pytest.raises(KeyError):
        phones_to_ids(phones, {"HH": 2})


def test_transformer_tts_recipe_id_mapping():
    """examples/transformer_tts/ljspeech/synthesize_e2e.py:84-90: punctuation kept, unknown phones -> ','."""
    from parakeet_amd.frontend import English, phones_to_ids_transformer_tts
    fe = English()
    phones = fe.phoneticize("Hello, world!")
    table = {p: i for i, p in enumerate(["<pad>", "<unk>", ",", "!", "HH", "AH0", "L", "OW1", "W", "ER1", "D", "<eos>"])}
    ids = phones_to_ids_transformer_tts(phones, table)
    inner = [p for p in phones[1:-1] if not p.isspace()]
    assert len(ids) == len(inner) and ids.dtype.kind == "i"
    assert [table.get(p, table[","]) for p in inner] == ids.tolist()
    assert table["!"] in ids.tolist() and table[","] in ids.tolist()          # punctuation survives


def test_vocab_symbol_names_are_plain_attributes_like_the_reference():
    """parakeet/frontend/vocab.py:57-60 stores the four special symbols as instance attributes: code that re-assigns them
    (or reads them after changing the table) must work."""
    from parakeet_amd.frontend.vocab import Vocab
    v = Vocab(["a", "b"])
    assert (v.padding_symbol, v.unk_symbol, v.start_symbol, v.end_symbol) == ("<pad>", "<unk>", "<s>", "</s>")
    assert (v.padding_index, v.unk_index, v.start_index, v.end_index) == (0, 1, 2, 3) and v.num_specials == 4
    v.unk_symbol = "b"
    assert v.unk_index == v.lookup("b") == 5
    v.end_symbol = "missing"
    assert v.end_index == -1
    w = Vocab(["x"], start_symbol=None, end_symbol=None)
    assert w.start_index == -1 and w.end_index == -1 and len(w) == 3
