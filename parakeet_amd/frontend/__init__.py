"""Text frontends of the synthesis recipes: text -> phones -> ids, host side.

Mirrors ``parakeet/frontend`` of the reference: ``Vocab`` (vocab.py:20-130), ``get_punctuations``
(punctuation.py:16-36), ``normalize`` / ``normalize_numbers`` (normalizer/normalizer.py:22-34,
normalizer/numbers.py:17-86), ``English`` / ``EnglishCharacter`` (phonectic.py:44-210) and the id mapping of the
synthesis recipe (examples/fastspeech2/ljspeech/synthesize_e2e.py:88-96).

The reference delegates grapheme-to-phoneme conversion to the third-party ``g2p_en`` package (CMUdict lookup, a POS
tagger for heteronyms, a neural predictor for unknown words) and number spelling to ``inflect``; neither is available
offline.  ``LexiconG2p`` stands in for ``g2p_en.G2p`` with the same call contract -- text in, a flat list of ARPAbet
phones with " " between words and punctuation marks as their own tokens out -- driven by a pronunciation lexicon in
CMUdict format that the caller provides (``lexicon=`` path; a small demonstration lexicon ships with the package),
with letter-to-sound rules for words the lexicon lacks.

Mandarin (``zh_frontend.Frontend``, zh_frontend.py:30-254 with tone_sandhi.py and zh_normalization/): the reference's
own logic -- text normalisation, merge rules, tone sandhi, initial / final splitting, erhua, id mapping -- over ONE
caller-supplied resource, a pinyin lexicon (``PinyinLexicon``), in place of the jieba / pypinyin / g2pM dictionaries;
pinned against the reference source run over dictionary stand-ins (tools/make_golden_zh.py).  ``ParakeetPinyin`` /
``ParakeetPinyinWithTone`` (pinyin.py:55-215): the pinyin phonologies of the Mandarin Tacotron2 recipes, same lexicon.
"""
from .vocab import Vocab
from .punctuation import get_punctuations
from .normalizer import normalize, normalize_numbers, full2half_width, half2full_width
from .g2p import LexiconG2p, ARPABET_PHONEMES
from .phonectic import ARPABET, ARPABETWithStress, Chinese, English, EnglishCharacter, Phonetics
from .phone_map import CachedTextToIds, phones_to_ids, phones_to_ids_transformer_tts, read_phone_id_map, text_to_ids
from .zh_frontend import Frontend, PinyinLexicon
from .zh_normalization import TextNormalizer
from .tone_sandhi import ToneSandhi
from .pinyin import ParakeetPinyin, ParakeetPinyinWithTone
from .generate_lexicon import generate_lexicon

__all__ = ["Vocab", "get_punctuations", "normalize", "normalize_numbers", "full2half_width", "half2full_width",
           "LexiconG2p", "ARPABET_PHONEMES", "ARPABET", "ARPABETWithStress", "English", "EnglishCharacter", "Phonetics", "phones_to_ids",
           "read_phone_id_map", "text_to_ids", "phones_to_ids_transformer_tts", "Frontend", "PinyinLexicon", "TextNormalizer", "ToneSandhi",
           "ParakeetPinyin", "ParakeetPinyinWithTone", "generate_lexicon", "Chinese", "CachedTextToIds"]
