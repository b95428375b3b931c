"""Mandarin frontend: text -> phones / tones -> ids, with the interface of parakeet/frontend/zh_frontend.py
(``Frontend`` :30-254: ``get_phonemes``, ``get_input_ids``) -- what the baker / aishell3 recipes call before
FastSpeech2 and SpeedySpeech (examples/speedyspeech/baker/synthesize_e2e.py:113-121).

Both of the reference's pipelines are here: ``g2p_model="pypinyin"`` (initial / final styles) and ``"g2pM"`` (tone-number
syllables split with the table of generate_lexicon.py, :78-92).

The reference leans on three packages that cannot be installed here and whose dictionaries are their substance:
jieba (word segmentation + part-of-speech tags), pypinyin (characters -> pinyin) and g2pM.  This module keeps every
piece of the reference's OWN logic -- text normalisation (zh_normalization.py), the merge rules and tone sandhi
(tone_sandhi.py), initial / final splitting with the i / ii / iii distinction, erhua merging, "sp" at sentence ends,
phone / tone id mapping with the unknown -> "sp" / "0" fall-backs and the split of merged erhua finals -- and replaces
the dictionaries by ONE resource the caller supplies: a pinyin lexicon (``PinyinLexicon``: word, syllables, optional
part of speech).  Segmentation is forward maximum matching over that lexicon.  The package ships a demonstration
lexicon of a few hundred entries (data/zh_demo_lexicon.txt); a deployment generates a full one once, on a machine that
has pypinyin + jieba, and passes its path.  Characters the lexicon lacks are reported in ``Frontend.missing`` and
come out as "sp" through the recipe's unknown-phone rule (:176-181).

Host-side text processing; no device work, no torch.
"""
import os
import re

import numpy as np

from .pinyin_split import split_syllable
from .tone_sandhi import ToneSandhi
from .zh_normalization import TextNormalizer

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_ASCII_WORD = re.compile(r"[A-Za-z]+")


class PinyinLexicon:
    """word -> (syllables, pos).  File format: ``word syl [syl ...] [#pos]``, '#' lines are comments."""

    def __init__(self, path=None, entries=None):
        self.words = {}
        if path is None and entries is None:
            path = os.path.join(_DATA, "zh_demo_lexicon.txt")
        if path is not None:
            with open(path, "rt", encoding="utf-8") as f:
                for line in f:
                    parts = line.split()
                    if not parts or parts[0].startswith("#"):
                        continue
                    pos = None
                    if parts[-1].startswith("#"):
                        pos, parts = parts[-1][1:], parts[:-1]
                    word, syl = parts[0], tuple(parts[1:])
                    if len(syl) != len(word):
                        raise ValueError(f"lexicon entry {word!r}: {len(syl)} syllables for {len(word)} characters")
                    self.words[word] = (syl, pos or ("n" if len(word) > 1 else "x"))
        for word, (syl, pos) in (entries or {}).items():
            self.words[word] = (tuple(syl), pos)
        self.max_len = max((len(w) for w in self.words), default=1)

    def segment(self, sentence):
        """[(word, pos)]: forward maximum matching; runs of ASCII letters are 'eng' (skipped by the frontend like jieba's
        tag, zh_frontend.py:108-109); anything unknown is a single character tagged 'x'."""
        out, i, n = [], 0, len(sentence)
        while i < n:
            m = _ASCII_WORD.match(sentence, i)
            if m:
                out.append((m.group(0), "eng"))
                i = m.end()
                continue
            if sentence[i].isspace():
                i += 1
                continue
            for L in range(min(self.max_len, n - i), 0, -1):
                w = sentence[i:i + L]
                if w in self.words:
                    out.append((w, self.words[w][1]))
                    i += L
                    break
            else:
                out.append((sentence[i], "x"))
                i += 1
        return out

    def pinyin(self, word, missing=None):
        """One syllable per character (tone-number pinyin); an unknown character is passed through, like pypinyin does
        with what it cannot convert."""
        if word in self.words:
            return list(self.words[word][0])
        out = []
        for piece, _ in self.segment(word):
            if piece in self.words:
                out += list(self.words[piece][0])
            else:
                out += list(piece)
                if missing is not None and "㐀" <= piece <= "鿿":
                    missing.append(piece)
        return out

    def cut_for_search(self, word):
        """jieba.cut_for_search on a single word: its dictionary 2-grams, then 3-grams, then the word itself."""
        out = []
        if len(word) > 2:
            out += [word[i:i + 2] for i in range(len(word) - 1) if word[i:i + 2] in self.words]
        if len(word) > 3:
            out += [word[i:i + 3] for i in range(len(word) - 2) if word[i:i + 3] in self.words]
        return out + [word]


class Frontend:
    def __init__(self, g2p_model="pypinyin", phone_vocab_path=None, tone_vocab_path=None, lexicon=None,
                 neutral_words=None):
        if g2p_model not in ("pypinyin", "g2pM"):
            raise ValueError(f"g2p_model {g2p_model!r}: pypinyin or g2pM")
        # "g2pM": the reference asks the g2pM network for tone-number syllables and splits them with the table of
        # generate_lexicon (:40-44, :78-92); here the lexicon answers in the network's place, the table path is the same
        self.g2p_model = g2p_model
        if g2p_model == "g2pM":
            from .generate_lexicon import generate_lexicon
            self.pinyin2phone = generate_lexicon(with_tone=True, with_erhua=False)
        self.lexicon = lexicon if isinstance(lexicon, PinyinLexicon) else PinyinLexicon(lexicon)
        self.missing = []
        kw = {} if neutral_words is None else {"neutral_words": neutral_words}
        self.tone_modifier = ToneSandhi(self.lexicon.cut_for_search, self._finals_of, **kw)
        self.text_normalizer = TextNormalizer()
        self.punc = "：，；。？！“”‘’':,;.?!"
        # words whose 儿 is (not) a rhotic suffix (:42-51); short lists, extend per deployment
        self.must_erhua = {"小院儿", "胡同儿", "范儿", "老汉儿", "撒欢儿"}
        self.not_erhua = {"女儿", "男儿", "婴儿", "幼儿", "孤儿", "妻儿", "孙儿", "侄儿", "花儿", "鸟儿", "马儿", "猫儿", "狗儿", "虫儿"}
        self.vocab_phones, self.vocab_tones = {}, {}
        for path, table in ((phone_vocab_path, self.vocab_phones), (tone_vocab_path, self.vocab_tones)):
            if path:
                with open(path, "rt", encoding="utf-8") as f:
                    for line in f:
                        if line.strip():
                            key, idx = line.strip().split()
                            table[key] = int(idx)

    # ---- pinyin
    def _finals_of(self, word):
        return [split_syllable(s)[1] for s in self.lexicon.pinyin(word)]

    def _get_initials_finals(self, word):
        """(:63-92) with the i -> ii / iii distinction after z c s / zh ch sh r."""
        initials, finals = [], []
        if self.g2p_model == "g2pM":
            for syl in self.lexicon.pinyin(word, self.missing):
                syl = syl.replace("u:", "v")
                if syl in self.pinyin2phone:
                    c, v = self.pinyin2phone[syl].split(" ")
                else:                      # not pinyin (punctuation, an unread character): passed through (:89-92)
                    c = v = syl
                initials.append(c)
                finals.append(v)
            return initials, finals
        for syl in self.lexicon.pinyin(word, self.missing):
            c, v = split_syllable(syl)
            if re.match(r"i\d", v):
                if c in ("z", "c", "s"):
                    v = re.sub("i", "ii", v)
                elif c in ("zh", "ch", "sh", "r"):
                    v = re.sub("i", "iii", v)
            initials.append(c)
            finals.append(v)
        return initials, finals

    def _merge_erhua(self, initials, finals, word, pos):
        """(:142-160): a final 儿 read er2 / er5 becomes an 'r' on the previous final (before its tone digit)."""
        if word not in self.must_erhua and (word in self.not_erhua or pos in {"a", "j", "nr"}):
            return initials, finals
        new_i, new_f = [], []
        assert len(finals) == len(word)
        for i, phn in enumerate(finals):
            if i == len(finals) - 1 and word[i] == "儿" and phn in {"er2", "er5"} and word[-2:] not in self.not_erhua and new_f:
                new_f[-1] = new_f[-1][:-1] + "r" + new_f[-1][-1]
            else:
                new_f.append(phn)
                new_i.append(initials[i])
        return new_i, new_f

    def _g2p(self, sentences, merge_sentences=True, with_erhua=True):
        """(:95-140)."""
        phones_list = []
        for seg in sentences:
            phones, initials, finals = [], [], []
            seg_cut = self.tone_modifier.pre_merge_for_modify(self.lexicon.segment(seg))
            for word, pos in seg_cut:
                if pos == "eng":
                    continue
                sub_i, sub_f = self._get_initials_finals(word)
                sub_f = self.tone_modifier.modified_tone(word, pos, sub_f)
                if with_erhua:
                    sub_i, sub_f = self._merge_erhua(sub_i, sub_f, word, pos)
                initials += sub_i
                finals += sub_f
            for c, v in zip(initials, finals):
                if c and c not in self.punc:
                    phones.append(c)
                if v and v not in self.punc:
                    phones.append(v)
            if initials and initials[-1] in self.punc:      # sp between sentences, in place of the last punctuation
                phones.append("sp")
            phones_list.append(phones)
        if merge_sentences:
            phones_list = [sum(phones_list, [])]
        return phones_list

    # ---- ids
    def _p2id(self, phonemes):
        return np.array([self.vocab_phones[p if p in self.vocab_phones else "sp"] for p in phonemes], np.int64)

    def _t2id(self, tones):
        return np.array([self.vocab_tones[t if t in self.vocab_tones else "0"] for t in tones], np.int64)

    def _get_phone_tone(self, phonemes, get_tone_ids=False):
        """(:176-217): phones (and tones); a merged erhua final the vocabulary lacks is split back into final + er."""
        phones, tones = [], []
        if get_tone_ids and self.vocab_tones:
            for full in phonemes:
                m = re.match(r"^(\w+)([012345])$", full)
                if not m:
                    phones.append(full)
                    tones.append("0")
                    continue
                phone, tone = m.group(1), m.group(2)
                if len(phone) >= 2 and phone != "er" and phone[-1] == "r" and phone not in self.vocab_phones and \
                        phone[:-1] in self.vocab_phones:
                    phones += [phone[:-1], "er"]
                    tones += [tone, "2"]
                else:
                    phones.append(phone)
                    tones.append(tone)
        else:
            for phone in phonemes:
                if len(phone) >= 3 and phone[:-1] != "er" and phone[-2] == "r" and phone not in self.vocab_phones and \
                        (phone[:-2] + phone[-1]) in self.vocab_phones:
                    phones += [phone[:-2] + phone[-1], "er2"]
                else:
                    phones.append(phone)
        return phones, tones

    def get_phonemes(self, sentence, merge_sentences=True, with_erhua=True):
        return self._g2p(self.text_normalizer.normalize(sentence), merge_sentences=merge_sentences, with_erhua=with_erhua)

    def get_input_ids(self, sentence, merge_sentences=True, get_tone_ids=False):
        """{"phone_ids": [int64 array per part], "tone_ids": [...]} (:228-254; the reference wraps them in tensors)."""
        result, phone_parts, tone_parts = {}, [], []
        for part in self.get_phonemes(sentence, merge_sentences=merge_sentences):
            phones, tones = self._get_phone_tone(part, get_tone_ids=get_tone_ids)
            if tones:
                tone_parts.append(self._t2id(tones))
            if phones:
                phone_parts.append(self._p2id(phones))
        if tone_parts:
            result["tone_ids"] = tone_parts
        if phone_parts:
            result["phone_ids"] = phone_parts
        return result
