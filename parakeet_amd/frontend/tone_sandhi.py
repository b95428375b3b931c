"""Mandarin tone sandhi on (word, part-of-speech, finals) triples: the rules of parakeet/frontend/tone_sandhi.py
(``ToneSandhi.pre_merge_for_modify`` :331-338 and ``modified_tone`` :340-343), restated over two hooks instead of the
jieba / pypinyin calls the reference makes:

    cut_for_search(word) -> sub-words  (jieba.cut_for_search, used to split a word in two, :143-154)
    finals_of(word)      -> finals with tone digits (pypinyin lazy_pinyin(..., FINALS_TONE3), :282-285, :309-312)

Rules (finals carry the tone as their last character; "5" = neutral):
  不   "X不Y" (3 characters): 不 neutral; otherwise 不 before a 4th tone -> 2nd tone                       (:103-114)
  一   inside digit strings unchanged; "X一X": neutral; "第一": 1st tone; before a 4th tone -> 2nd, else 4th  (:116-139)
  neutral tones: reduplicated n. / v. / a.; sentence-final particles; 的 地 得; aspect 了 着 过; plural / nominal
       们 子; locatives 上 下 里; directional 来 去; the measure word 个; and a word list                    (:67-101)
  third tone: 33 -> 23; 333 by the word's internal split (2+1: 223, 1+2: 323); 4-character idioms by halves (:156-195)
The neutral-tone word list of the reference (~400 lexical entries) is data, not logic: ``NEUTRAL_WORDS`` below is a
short list of very common ones, and ``ToneSandhi(neutral_words=...)`` takes the full set a deployment wants.
"""

NEUTRAL_WORDS = frozenset("""
东西 朋友 什么 怎么 这么 那么 这个 那个 时候 喜欢 先生 告诉 漂亮 明白 清楚 地方 事情 意思 部分 关系 学生 认识 消息 休息
衣服 头发 太阳 月亮 耳朵 眼睛 豆腐 玻璃 葡萄 萝卜 窗户 钥匙 苍蝇 蘑菇 骆驼 石头 木头 舌头 骨头 馒头 枕头 拳头 指头 后头 前头
上头 里头 丈夫 大夫 姑娘 媳妇 丫头 兄弟 亲戚 伙计 客气 脾气 力气 运气 福气 便宜 厉害 麻烦 热闹 舒服 凉快 暖和 结实 老实
聪明 糊涂 马虎 讲究 功夫 工夫 故事 买卖 生意 主意 行李 收拾 打算 打听 打扮 商量 答应 招呼 称呼 笑话 耽误 咳嗽 哆嗦 吓唬
多少 多么 人家 名字 知识 精神 记性 见识 本事 学问 状元 师傅 师父 护士 和尚 喇叭 灯笼 风筝 胳膊 屁股 尾巴 嘴巴 下巴 哑巴
""".split())

NOT_NEUTRAL_ZI = frozenset("男子 女子 分子 原子 量子 莲子 石子 瓜子 电子".split())
_PARTICLES = "吧呢哈啊呐噻嘛吖嗨呐哦哒额滴哩哟喽啰耶喔诶"


def _set_tone(final, tone):
    return final[:-1] + tone


def _all_third(finals):
    return all(f[-1] == "3" for f in finals)


class ToneSandhi:
    def __init__(self, cut_for_search, finals_of, neutral_words=NEUTRAL_WORDS):
        self._cut, self._finals_of = cut_for_search, finals_of
        self.must_neural_tone_words = set(neutral_words)
        self.must_not_neural_tone_words = set(NOT_NEUTRAL_ZI)

    # ---------------------------------------------------------------- per-word tone changes
    def split_word(self, word):
        """Two parts of a word around its shortest dictionary sub-word (:143-154)."""
        subs = sorted(self._cut(word), key=len)
        first = subs[0]
        if word.find(first) == 0:
            return [first, word[len(first):]]
        return [word[:-len(first)], first]

    def _bu(self, word, finals):
        if len(word) == 3 and word[1] == "不":
            finals[1] = _set_tone(finals[1], "5")
            return finals
        for i, ch in enumerate(word):
            if ch == "不" and i + 1 < len(word) and finals[i + 1][-1] == "4":
                finals[i] = _set_tone(finals[i], "2")
        return finals

    def _yi(self, word, finals):
        if "一" in word and all(c.isnumeric() for c in word if c != "一"):
            return finals                                   # 一 inside a number sequence
        if len(word) == 3 and word[1] == "一" and word[0] == word[2]:
            finals[1] = _set_tone(finals[1], "5")            # 看一看
        elif word.startswith("第一"):
            finals[1] = _set_tone(finals[1], "1")
        else:
            for i, ch in enumerate(word):
                if ch == "一" and i + 1 < len(word):
                    finals[i] = _set_tone(finals[i], "2" if finals[i + 1][-1] == "4" else "4")
        return finals

    def _listed(self, w):
        return w in self.must_neural_tone_words or w[-2:] in self.must_neural_tone_words

    def _neutral(self, word, pos, finals):
        for j in range(1, len(word)):
            if word[j] == word[j - 1] and pos[0] in "nva":
                finals[j] = _set_tone(finals[j], "5")        # 奶奶, 试试
        ge = word.find("个")
        last = word[-1] if word else ""
        if last and last in _PARTICLES:
            finals[-1] = _set_tone(finals[-1], "5")
        elif last and last in "的地得":
            finals[-1] = _set_tone(finals[-1], "5")
        elif len(word) == 1 and word in "了着过" and pos in ("ul", "uz", "ug"):
            finals[-1] = _set_tone(finals[-1], "5")
        elif len(word) > 1 and last in "们子" and pos in ("r", "n") and word not in self.must_not_neural_tone_words:
            finals[-1] = _set_tone(finals[-1], "5")
        elif len(word) > 1 and last in "上下里" and pos in ("s", "l", "f"):
            finals[-1] = _set_tone(finals[-1], "5")
        elif len(word) > 1 and last in "来去" and word[-2] in "上下进出回过起开":
            finals[-1] = _set_tone(finals[-1], "5")
        elif (ge >= 1 and (word[ge - 1].isnumeric() or word[ge - 1] in "几有两半多各整每做是")) or word == "个":
            finals[ge] = _set_tone(finals[ge], "5")
        elif self._listed(word):
            finals[-1] = _set_tone(finals[-1], "5")
        parts = self.split_word(word)
        cut = len(parts[0])
        halves = [finals[:cut], finals[cut:]]
        for part, half in zip(parts, halves):
            if self._listed(part):
                half[-1] = _set_tone(half[-1], "5")
        return halves[0] + halves[1]

    def _three(self, word, finals):
        if len(word) == 2 and _all_third(finals):
            finals[0] = _set_tone(finals[0], "2")
        elif len(word) == 3:
            parts = self.split_word(word)
            if _all_third(finals):
                if len(parts[0]) == 2:                       # 蒙古 / 包
                    finals[0], finals[1] = _set_tone(finals[0], "2"), _set_tone(finals[1], "2")
                elif len(parts[0]) == 1:                     # 纸 / 老虎
                    finals[1] = _set_tone(finals[1], "2")
            else:
                cut = len(parts[0])
                halves = [finals[:cut], finals[cut:]]
                for i, sub in enumerate(halves):
                    if _all_third(sub) and len(sub) == 2:    # 所有 / 人
                        halves[i][0] = _set_tone(halves[i][0], "2")
                    elif i == 1 and not _all_third(sub) and halves[i][0][-1] == "3" and halves[0][-1][-1] == "3":
                        halves[0][-1] = _set_tone(halves[0][-1], "2")   # 好 / 喜欢
                    finals = halves[0] + halves[1]
        elif len(word) == 4:
            out = []
            for sub in (finals[:2], finals[2:]):
                if _all_third(sub):
                    sub[0] = _set_tone(sub[0], "2")
                out += sub
            finals = out
        return finals

    def modified_tone(self, word, pos, finals):
        finals = self._bu(word, finals)
        finals = self._yi(word, finals)
        finals = self._neutral(word, pos, finals)
        return self._three(word, finals)

    # ---------------------------------------------------------------- word merging before the tone changes
    @staticmethod
    def _merge_bu(seg):
        out, last = [], ""
        for word, pos in seg:
            if last == "不":
                word = last + word
            if word != "不":
                out.append((word, pos))
            last = word
        if last == "不":
            out.append((last, "d"))
        return out

    @staticmethod
    def _merge_yi(seg):
        out = []
        for i, (word, pos) in enumerate(seg):                # 听 一 听 -> 听一听
            if i >= 1 and word == "一" and i + 1 < len(seg) and seg[i - 1][0] == seg[i + 1][0] and seg[i - 1][1] == "v":
                out[i - 1][0] = out[i - 1][0] + "一" + out[i - 1][0]
            elif i >= 2 and seg[i - 1][0] == "一" and seg[i - 2][0] == word and pos == "v":
                continue
            else:
                out.append([word, pos])
        merged = []
        for word, pos in out:                                # a lone 一 joins the word behind it
            if merged and merged[-1][0] == "一":
                merged[-1][0] = merged[-1][0] + word
            else:
                merged.append([word, pos])
        return merged

    @staticmethod
    def _merge_reduplication(seg):
        out = []
        for word, pos in seg:
            if out and word == out[-1][0]:
                out[-1][0] = out[-1][0] + word
            else:
                out.append([word, pos])
        return out

    def _merge_thirds(self, seg, whole_words):
        """Join a word to the one before when third tones meet (whole words :277-303, or just the touching syllables
        :308-328), unless the first is a reduplication or the pair would exceed 3 characters."""
        fin = [self._finals_of(w) for w, _ in seg]
        out, merged_last = [], [False] * len(seg)
        for i, (word, pos) in enumerate(seg):
            if whole_words:
                meet = i >= 1 and _all_third(fin[i - 1]) and _all_third(fin[i])
            else:
                meet = i >= 1 and fin[i - 1][-1][-1] == "3" and fin[i][0][-1] == "3"
            if meet and not merged_last[i - 1]:
                prev = seg[i - 1][0]
                if not (len(prev) == 2 and prev[0] == prev[1]) and len(prev) + len(word) <= 3:
                    out[-1][0] = out[-1][0] + word
                    merged_last[i] = True
                    continue
            out.append([word, pos])
        return out

    @staticmethod
    def _merge_er(seg):
        out = []
        for i, (word, pos) in enumerate(seg):
            if i >= 1 and word == "儿":
                out[-1][0] = out[-1][0] + word
            else:
                out.append([word, pos])
        return out

    def pre_merge_for_modify(self, seg):
        seg = self._merge_bu(seg)
        seg = self._merge_yi(seg)
        seg = self._merge_reduplication(seg)
        seg = self._merge_thirds(seg, True)
        seg = self._merge_thirds(seg, False)
        return self._merge_er(seg)
