"""CPU restatement of the reference's text front end (lexicon parser and pronouncer).

TEST INFRASTRUCTURE ONLY: importable from tests/.  Nothing under speechrecognition_amd/ may import this module.
Parity status: UNPINNED -- the reference's tests hold no golden pronunciations and the reference cannot be built
here; this file restates src/pronounce.cpp statement by statement, and the expected phone
strings in tests/golden/lexicon/expected.txt were derived by hand from that source.

Each function cites the reference lines it follows.  Where the reference has no defined behaviour (falls off the
end of a non-void function, indexes an empty string) this restatement raises Undefined, so a test can tell "the
reference gives up" (Unpronounceable, its `throw 1`) from "the reference does not say".
"""

NAMES = ["_", "AA", "AE", "AH", "AO", "AW", "AY", "B", "CH", "D", "DH", "EH", "ER", "EY", "F", "G", "HH", "IH", "IY",
         "JH", "K", "L", "M", "N", "NG", "OW", "OY", "P", "R", "S", "SH", "T", "TH", "UH", "UW", "V", "W", "Y", "Z",
         "ZH"]                                                                           # src/pronounce.hpp:30-33
SIL = 0


class Unpronounceable(Exception):
    """src/pronounce.cpp:253-254: "Warning! Cannot pronounce" followed by throw 1."""


class Undefined(Exception):
    """The reference's behaviour is undefined for this input."""


def find_phone(w):                                                                       # src/pronounce.cpp:16-26
    for i, ch in enumerate(w):
        if ch in "0123456789":
            w = w[:i] + w[i + 1:]     # only the first digit is removed
            break
    for i, n in enumerate(NAMES):
        if w == n:
            return i
    raise Undefined("phone name %r" % w)


def read_lexicon(lines):                                                                 # src/pronounce.cpp:105-138
    """lines: iterable of text lines without their newline.  Returns {word: [pronunciation, ...]}."""
    pr = {}
    for buf in lines:
        if buf[:1] == ";":
            continue
        # `is >> word`, then `while (is.good()) { is >> ph; push(findPhone(ph)); }`: good() is still true after a
        # token that is followed by blanks, the next extraction fails and leaves ph as it was
        toks = buf.split()
        word = toks[0] if toks else ""
        paren = word.find("(")
        if paren != -1:
            word = word[:paren]
        phs = toks[1:]
        trailing_blank = len(buf) > 0 and buf[-1] in " \t" and len(toks) >= 1
        p = [find_phone(x) for x in phs]
        if trailing_blank:
            if not phs:
                raise Undefined("findPhone of an empty name")
            p.append(p[-1])
        pr.setdefault(word, []).append(p)
    return pr


class Pronouncer:
    def __init__(self, main, prefixes, suffixes):                                       # src/pronounce.cpp:150-156
        self.main, self.prefixes, self.suffixes = main, prefixes, suffixes

    def _get(self, w):                                                                   # src/pronounce.cpp:141-147
        return self.main.get(w)

    def pronounce_word(self, w):                                                         # src/pronounce.cpp:178-257
        if w == "":
            raise Undefined("w[w.length()-1] of an empty word")
        res = []
        if w[-1] in "-'":
            w = w[:-1]
        p = self._get(w)
        if p is not None:
            return [SIL] + p[0]                                                          # :183-185
        if len(w) < 4:
            raise Undefined("no return for a short word that is not in the lexicon")
        if "-" in w:                                                                     # :188-199
            first = True
            while "-" in w:
                idx = w.index("-")
                piece = self.pronounce_word(w[:idx])
                res += ([SIL] if first else []) + piece
                w = w[idx + 1:]
                first = False
            return res + [SIL] + self.pronounce_word(w)
        for suf in sorted(self.suffixes):                                                # :202-211 (std::map order)
            if len(w) >= len(suf) and w[len(w) - len(suf):] == suf:
                p = self._get(w[:len(w) - len(suf)])
                if p is not None:
                    return [SIL] + p[0]
        for pre in sorted(self.prefixes):                                                # :213-255
            if len(pre) < len(w) and w[:len(pre)] == pre:
                p = self._get(w[len(pre):])
                if p is not None:
                    return [SIL] + p[0]
                for suf in sorted(self.suffixes):
                    if len(w) - len(pre) >= len(suf) and w[len(w) - len(suf):] == suf:
                        p = self._get(w[len(pre):len(w) - len(suf)])
                        if p is not None:
                            return [SIL] + p[0]
            # still inside the prefix loop: the split search, then the give-up
            split = 0
            while split < len(w) // 2 - 1:
                for side in (-1, 1):
                    s = len(w) // 2 + side * split
                    p = self._get(w[:s])
                    if p is not None:
                        q = self._get(w[s:])
                        if q is not None:
                            # append_pron(res, p, true); append_pron(res, q, false): the pronlist overload ignores
                            # its third argument (:166-168), so both halves get a silence
                            return [SIL] + p[0] + [SIL] + q[0]
                    if split == 0:
                        break
                split += 1
            raise Unpronounceable(w)
        raise Undefined("no return with an empty prefix lexicon")

    def pronounce(self, s):                                                              # src/pronounce.cpp:260-278
        res = []
        words = s.split()
        if s == "" or s[-1:].isspace():
            # `while (is.good()) { is >> w; ... }` runs once more with an empty w
            raise Undefined("empty word at the end of the text")
        for w in words:
            res += self.pronounce_word(w)
        return res + [SIL]


def load(main_fn, pre_fn, suf_fn):
    def rd(fn):
        with open(fn, encoding="latin-1") as f:   # bytes, as the reference reads them
            return read_lexicon(f.read().splitlines())
    return Pronouncer(rd(main_fn), rd(pre_fn), rd(suf_fn))
