#!/usr/bin/env python
"""Write a pinyin lexicon for parakeet_amd.frontend.PinyinLexicon on a machine that HAS jieba and pypinyin (this image
does not; the script is the one-off export a deployment runs elsewhere).  Every word of jieba's dictionary (with its
part-of-speech tag) and every single CJK character pypinyin knows get one line:  ``word syl [syl ...] #pos`` -- tone-
number pinyin, 5 for the neutral tone, ü as v: exactly what the reference's frontend asks pypinyin for
(parakeet/frontend/zh_frontend.py:68-71).

usage: python tools/make_zh_lexicon.py out.txt [--max-words N]"""
import argparse
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--max-words", type=int, default=0, help="keep only the N most frequent dictionary words (0 = all)")
    args = ap.parse_args()
    try:
        import jieba
        import jieba.posseg  # noqa: F401  (loads the tagged dictionary)
        from pypinyin import Style, lazy_pinyin
        from pypinyin.constants import PINYIN_DICT
    except ImportError as e:
        sys.exit(f"make_zh_lexicon.py needs jieba and pypinyin: {e}")

    def syllables(word):
        return lazy_pinyin(word, neutral_tone_with_five=True, style=Style.TONE3, v_to_u=False)

    jieba.initialize()
    entries = []
    with open(jieba.get_dict_file().name, "rt", encoding="utf-8") as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 2:
                entries.append((parts[0], int(parts[1]), parts[2] if len(parts) > 2 else "n"))
    entries.sort(key=lambda e: -e[1])
    if args.max_words:
        entries = entries[:args.max_words]
    seen = set()
    with open(args.out, "wt", encoding="utf-8") as out:
        for word, _, pos in entries:
            syl = syllables(word)
            if len(syl) == len(word) and all(s[-1].isdigit() for s in syl):
                out.write(f"{word} {' '.join(syl)} #{pos}\n")
                seen.add(word)
        for code in sorted(PINYIN_DICT):
            ch = chr(code)
            if ch not in seen and "㐀" <= ch <= "鿿":
                syl = syllables(ch)
                if len(syl) == 1 and syl[0][-1].isdigit():
                    out.write(f"{ch} {syl[0]} #x\n")
    print(f"wrote {args.out}")


if __name__ == "__main__":
    main()
