"""Inputs shared by tools/make_golden_en.py (runs the reference's English frontend over stand-ins for ``inflect`` and
``g2p_en``) and tests/test_en_frontend_golden_cpu.py (replays them on parakeet_amd.frontend)."""

SENTENCES = [
    # plain text, punctuation, case, accents, characters outside the kept set
    "Hello, world!",
    "The quick brown fox jumps over the lazy dog.",
    "Printing, in the only sense with which we are at present concerned, differs from most if not from all the arts.",
    "Is this it? Yes... no - maybe!",
    "Café déjà vu, naïve résumé.",
    "She said: \"don't\" (twice); he didn't.",
    "It's the dog's bone, isn't it?",
    "A well-known state-of-the-art method.",
    "WAIT!!! What?? Really...",
    "tabs\tand\nnewlines   and   spaces",
    "i.e. that one, e.g. this one.",
    "",
    "   ",
    "?!",
    # cardinals
    "0", "7", "10", "13", "20", "21", "99", "100", "101", "110", "999",
    "I have 3 books and 15 pens.",
    "1000", "1001", "3000", "3001", "10000", "12345", "100000", "1000000", "1234567", "1000000000",
    "There were 4096 of them, or 65536.",
    # thousands separators
    "12,345 people", "1,000,000 stars", "7,000", "1,2", "in 1,984 cases",
    # the year branch 1000 < n < 3000
    "1984", "1905", "1900", "2000", "2001", "2007", "2009", "2010", "2011", "2021", "1066", "1100", "1101", "2999", "1010",
    "In 1776 and in 1999.",
    # ordinals
    "1st", "2nd", "3rd", "4th", "5th", "8th", "9th", "11th", "12th", "13th", "20th", "21st", "22nd", "23rd", "30th",
    "40th", "99th", "100th", "101st", "111th", "112th", "120th", "1000th", "1001st", "the 3rd of May",
    # decimals
    "3.14", "0.5", "10.25 percent", "version 1.2.3", "2.0",
    # currency
    "$1", "$2", "$0.01", "$0.50", "$1.01", "$3.50", "$1,000", "$1,000.50", "$12.5.6", "$.5", "$0", "£20", "£1,500",
    "It costs $5 or £3.",
    # mixtures
    "Call 911 at 5pm on the 4th, it's $20.",
    "Room 101, 2nd floor, 12:30.",
    "He ran 26.2 miles in 1984 for $1,000,000.",
    "Zorblat quux 42!",
]

# sentences for the recipe's id mapping (examples/fastspeech2/ljspeech/synthesize_e2e.py:88-98: drop start / end, drop blanks,
# out-of-map symbols and punctuation -> "sp")
RECIPE_SENTENCES = ["Hello, world!", "The 3rd dog cost $5.", "Zorblat - quux?"]
