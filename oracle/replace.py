"""ORACLE (test infrastructure, not product): restatement of the reference's runtime Replace path --
`replace.Parse` / `ValidateAndResolve` (/root/reference/replace/template.go:45-291) and the emitted
ReplaceAllBytesAppend / ReplaceFirstBytes loops (/root/reference/internal/compiler/replace.go:205-323, 325-363,
393-453).  Pinned by tests/golden/replace_kats.json: the literal vectors of replace/template_test.go.

The emitted loop re-slices the input after every match and asks FindBytesReuse for the first match of the remainder:
  Q1   FindBytesReuse restarts at failure offset + 1 (misses some matches; same quirk as FindReader),
  Q4'  the match position is recovered by bytes.Index(remaining, match.Match) (replace.go:246): an earlier occurrence
       of the same text moves the splice point,
  Q12  the remainder begins a new "text": `^`, `\\b`, `(?m)^` see no byte before each match end.
`replace_all(..., quirks=True)` reproduces all three; `quirks=False` is the same loop over true leftmost-first matches
in their real context (what the GPU path computes): the FindAllBytes matches plus one attempt at offset len(input).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

from . import engines as E

LIT, FULL, IDX, NAME = "lit", "full", "idx", "name"


class TemplateError(ValueError):
    pass


def _is_letter(ch: str) -> bool:
    return ch.isalpha()


def _is_name_start(ch: str) -> bool:  # template.go:231-233
    return ch == "_" or _is_letter(ch)


def _is_name_continue(ch: str) -> bool:  # template.go:235-237 (unicode.IsDigit = category Nd)
    import unicodedata
    return ch == "_" or _is_letter(ch) or unicodedata.category(ch) == "Nd"


def parse(template: str) -> List[dict]:
    """replace.Parse (template.go:45-148).  The scan is over BYTES of the UTF-8 template (Go indexes a string by byte
    and converts single bytes to runes), so a non-ASCII byte after `$` is tested as the rune U+00xx."""
    t = template.encode("utf-8")
    segs: List[dict] = []
    i = 0
    lit0 = 0
    n = len(t)

    def flush(upto):
        if upto > lit0:
            segs.append({"type": LIT, "literal": t[lit0:upto].decode("utf-8", "surrogateescape")})

    while i < n:
        if t[i] != 0x24:
            i += 1
            continue
        flush(i)
        if i + 1 >= n:
            segs.append({"type": LIT, "literal": "$"})
            i += 1
            lit0 = i
            continue
        nxt = t[i + 1]
        if nxt == 0x24:
            segs.append({"type": LIT, "literal": "$"})
            i += 2
        elif nxt == 0x7B:  # ${...}  (template.go:151-196)
            close = t.find(b"}", i)
            if close == -1:
                raise TemplateError("at position %d: unclosed ${" % i)
            content = t[i + 2:close]
            if len(content) == 0:
                raise TemplateError("at position %d: empty ${}" % i)
            if 0x30 <= content[0] <= 0x39:
                if any(c < 0x30 or c > 0x39 for c in content):
                    raise TemplateError("at position %d: invalid capture reference: mixed digits and non-digits" % i)
                idx = int(content)
                segs.append({"type": FULL} if idx == 0 else {"type": IDX, "index": idx})
            else:
                name = content.decode("utf-8", "replace")   # isValidIdentifier ranges over RUNES of the content
                if not name or not _is_name_start(name[0]) or not all(_is_name_continue(c) for c in name[1:]):
                    raise TemplateError("at position %d: invalid capture name" % i)
                segs.append({"type": NAME, "name": name})
            i = close + 1
        elif nxt == 0x30:
            segs.append({"type": FULL})
            i += 2
        elif 0x31 <= nxt <= 0x39:  # template.go:199-213
            idx = nxt - 0x30
            used = 2
            if i + 2 < n and 0x30 <= t[i + 2] <= 0x39:
                idx = idx * 10 + (t[i + 2] - 0x30)
                used = 3
            segs.append({"type": IDX, "index": idx})
            i += used
        elif _is_name_start(chr(nxt)):  # rune(byte): bytes >= 0x80 are tested as U+0080..U+00FF
            end = i + 2
            while end < n and _is_name_continue(chr(t[end])):
                end += 1
            segs.append({"type": NAME, "name": t[i + 1:end].decode("latin-1") if any(c >= 0x80 for c in t[i + 1:end])
                         else t[i + 1:end].decode("ascii")})
            i = end
        else:
            segs.append({"type": LIT, "literal": "$"})
            i += 1
        lit0 = i
    flush(i)
    return segs


def validate_and_resolve(segs: List[dict], capture_names: Dict[str, int], num_captures: int) -> List[dict]:
    """Template.ValidateAndResolve (template.go:262-291): used by the precompiled variants; raises on a bad reference."""
    out = []
    for s in segs:
        if s["type"] == IDX:
            if s["index"] > num_captures:
                raise TemplateError("capture group %d out of range" % s["index"])
            out.append(s)
        elif s["type"] == NAME:
            if s["name"] not in capture_names:
                raise TemplateError("capture group %r not found" % s["name"])
            out.append({"type": IDX, "index": capture_names[s["name"]]})
        else:
            out.append(s)
    return out


def _expand(segs: List[dict], text: bytes, caps: List[int], names: Dict[str, int], ngroups: int) -> bytes:
    """generateTemplateExpansionBytes (replace.go:393-453): unknown names and out-of-range indices expand to nothing."""
    out = bytearray()
    for s in segs:
        ty = s["type"]
        if ty == LIT:
            out += s["literal"].encode("utf-8", "surrogateescape")
        elif ty == FULL:
            out += text[caps[0]:caps[1]]
        else:
            g = s["index"] if ty == IDX else names.get(s["name"], -1)
            if 1 <= g <= ngroups:
                out += text[caps[2 * g]:caps[2 * g + 1]]
    return bytes(out)


def replace_all(c: "E.Compiled", inp: bytes, template: str, quirks: bool = False, first_only: bool = False) -> bytes:
    segs = parse(template)
    from . import syntax as S
    names = {n: i for i, n in enumerate(S.capture_names(c.ast)) if i >= 1 and n}
    ngroups = c.prog.numcap // 2 - 1
    if not quirks:
        ms = c.find_machine.find_all(inp)
        caps_end = [0] * c.prog.numcap
        caps_end[0] = len(inp)
        ok, end = c.find_machine._attempt(inp, len(inp), len(inp), caps_end, set() if c.find_machine.memo else None)
        # the loop's FindBytesReuse also tries at offset len(input) (find.go:545-569); FindAllBytes does not (find.go:209-211).
        # An anchored pattern only ever tries offset 0 of the ORIGINAL input in the quirk-free reading.
        if ok and (not c.find_machine.anchored or len(inp) == 0):
            caps_end[1] = end
            ms.append(caps_end)
        if first_only:
            ms = ms[:1]
        out = bytearray()
        last = 0
        for m in ms:
            out += inp[last:m[0]]
            out += _expand(segs, inp, m, names, ngroups)
            last = m[1]
        out += inp[last:]
        return bytes(out)
    # the emitted loop, quirks included
    out = bytearray()
    last_end = 0
    remaining = inp
    offset = 0
    # Tagged-DFA programs (compiler.go:137-153): FindBytesReuse is the Tagged DFA's (tdfa.go:831-1052), and `r` is ONE struct for the
    # whole loop (replace.go:216): a group whose start tag is unset is left untouched (tdfa.go:1031-1046), so its field still holds the
    # text the last match that set it gave it -- a slice of `input` -- and the zero struct's empty field before that.
    tdfa = getattr(c, "tdfa", None)
    held = [0] * c.prog.numcap          # what the struct's fields hold, as offsets into `inp`
    while True:
        if tdfa is not None:
            t = tdfa.find(remaining)
            if t is None:
                break
            for g in range(1, c.prog.numcap // 2):
                if t[2 * g] >= 0:
                    held[2 * g], held[2 * g + 1] = t[2 * g] + offset, t[2 * g + 1] + offset
            caps = [t[0], t[1]]
        else:
            caps = c.find_machine.find(remaining)
            if caps is None:
                break
        match = remaining[caps[0]:caps[1]]
        idx = remaining.find(match)
        if idx < 0:
            break
        ms, me = offset + idx, offset + idx + len(match)
        out += inp[last_end:ms]
        if tdfa is not None:
            out += _expand(segs, inp, [offset + caps[0], offset + caps[1]] + held[2:], names, ngroups)
        else:
            out += _expand(segs, remaining, caps, names, ngroups)
        last_end = me
        if first_only:
            break
        if len(match) > 0:
            remaining = inp[me:]
            offset = me
        elif me < len(inp):
            remaining = inp[me + 1:]
            offset = me + 1
        else:
            break
    out += inp[last_end:]
    return bytes(out)
