"""Punctuation inventories (parakeet/frontend/punctuation.py:16-36)."""
__all__ = ["get_punctuations"]

EN_PUNCT = [" ", "-", "...", ",", ".", "?", "!"]
CN_PUNCT = ["、", "，", "；", "：", "。", "？", "！"]


def get_punctuations(lang):
    if lang == "en":
        return EN_PUNCT
    if lang == "cn":
        return CN_PUNCT
    raise ValueError(f"language {lang} Not supported")
