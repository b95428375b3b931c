"""Inputs shared by tools/make_golden_zh.py (runs the reference's Mandarin frontend over dictionary stand-ins) and
tests/test_zh_frontend_cpu.py."""

NORMALIZE = [
    "2021年3月15日，气温-3°C。", "今天是2020/10/29，下午3:05:09开会！", "他有1/3的股份，涨了12.5%；电话13812345678或010-62345678。",
    "价格在3.5~10.25元之间", "共10086个字，-10分，温度20度，编号00078，小数.22和3.20", "买了12多本书，第305页", "1999年", "98年12月",
    "摄氏度测试：38.5摄氏度", "一共100万元，5千米，23:59", "今天2月14号", "比分3-2", "电话+8613912345678", "1.50",
    "他说：“好的！”然后走了", "2000-01-01是千禧年", "-0.5和0.05", "第1名，第2个，3天，4月5日", "100000000人，20003元，1010个",
    "7:00到8:30", "零下-15.5℃",
]

# (word, part of speech, finals before sandhi): the reference's docstring examples and one case per rule
SANDHI = [
    ("家里", "s", ["ia1", "i3"]), ("看不懂", "v", ["an4", "u4", "ong3"]), ("不怕", "d", ["u4", "a4"]), ("不好", "d", ["u4", "ao3"]),
    ("一段", "m", ["i1", "uan4"]), ("一天", "m", ["i1", "ian1"]), ("第一", "m", ["i4", "i1"]), ("看一看", "v", ["an4", "i1", "an4"]),
    ("一零零", "m", ["i1", "ing2", "ing2"]), ("奶奶", "n", ["ai3", "ai3"]), ("看看", "v", ["an4", "an4"]), ("好吧", "y", ["ao3", "a5"]),
    ("我的", "r", ["uo3", "e5"]), ("了", "ul", ["e5"]), ("我们", "r", ["uo3", "en5"]), ("桌子", "n", ["uo1", "i3"]),
    ("男子", "n", ["an2", "i3"]), ("上来", "v", ["ang4", "ai2"]), ("两个", "m", ["iang3", "e4"]), ("个", "q", ["e4"]),
    ("朋友", "n", ["eng2", "iou3"]), ("你好", "l", ["i3", "ao3"]), ("蒙古包", "n", ["eng3", "u3", "ao1"]),
    ("纸老虎", "n", ["iii3", "ao3", "u3"]), ("所有人", "n", ["uo3", "iou3", "en2"]), ("买手表", "v", ["ai3", "ou3", "iao3"]),
    ("展览馆", "n", ["an3", "an3", "uan3"]), ("很好", "a", ["en3", "ao3"]), ("水果", "n", ["uei3", "uo3"]),
    ("豆腐", "n", ["ou4", "u3"]), ("什么", "r", ["en2", "e5"]), ("胡同儿", "n", ["u2", "ong4", "er2"]),
]

SENTENCES = [
    "你好，我们今天去北京。", "他不怕，也不好说！", "看一看这个东西吧？", "我有一个朋友，他很喜欢纸老虎。", "蒙古包里有奶奶的桌子；",
    "2021年3月5日下午3:05开会", "小孩儿在胡同儿里看花儿，女儿也去", "他说hello我不懂", "所有人都买手表", "第一天一样一起走", "听一听，想想",
    "展览馆很好，水果也很好", "气温-3度，涨了12.5%", "未登录的字：龘",
]

PHONES = ["<pad>", "<unk>", "sp", "b", "p", "m", "f", "d", "t", "n", "l", "g", "k", "h", "j", "q", "x", "zh", "ch", "sh", "r", "z", "c", "s",
          "a", "ai", "an", "ang", "ao", "e", "ei", "en", "eng", "er", "o", "ong", "ou", "i", "ii", "iii", "ia", "ian", "iang", "iao", "ie",
          "in", "ing", "iong", "iou", "u", "ua", "uai", "uan", "uang", "uei", "uen", "ueng", "uo", "v", "van", "ve", "vn", "uar", "<eos>"]
# phones with tone digits for the no-tone-id path (speedyspeech uses separate tones; fastspeech2 baker uses toned finals)
PHONES = PHONES[:-1] + [f + t for f in PHONES[24:-2] for t in "12345"] + ["<eos>"]
TONES = ["<pad>", "<unk>", "0", "1", "2", "3", "4", "5"]

# English ARPABET phonologies (parakeet/frontend/arpabet.py) over the demonstration CMUdict-format lexicon
ARPABET_TEXTS = ["Hello, world! This is a test.", "The quick brown fox? Yes.", "unknownword zyx, forty two dollars"]

# frontend/pinyin.py (ParakeetPinyin / ParakeetPinyinWithTone): sentences of the lexicon, the four marks of its inventory,
# a run of letters (kept together, as pypinyin leaves it) and syllables that exercise every rewriting rule the demonstration
# lexicon can reach (bo -> buo, ong / iong, in / ing, un / ui / iu, zi / zhi, y / w glides, ju / qu / xu)
PINYIN_SENTENCES = ["你好，我们今天去北京。", "他不怕，也不好说！", "看一看这个东西吧？", "所有人都买手表", "我有一个朋友，他很喜欢纸老虎。",
                    "他说hello我不懂", "第一天一样一起走", "展览馆很好，水果也很好", "听一听，想想", "蒙古包里有奶奶的桌子"]
