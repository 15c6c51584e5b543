"""ORACLE (test infrastructure, not product): restatement of the reference's Tagged-DFA capture engine
(/root/reference/internal/compiler/tdfa.go).  The reference selects it for patterns with captures and nested
quantifiers when the construction stays under 500 states (compiler.go:137-153); URLCapture, TDFASemVer and the
ipv4 streaming pattern are emitted this way.

Restated: construction (tdfa.go:111-290: start states, worklist subset construction over PRIORITY-ORDERED NFA sets
with pending tag actions, common-prefix hoisting, EOT acceptance), tables (584-794), the table-driven find loop
(831-994: first start that yields any accept wins; LAST accept along the walk wins; non-ASCII byte aborts the
attempt), result construction (998-1052) and the FindAll wrapper (compiler.go:602-655).  Pinned by
tests/golden/tdfa_tables.json: the literal tables of the three checked-in TDFA matchers.

Behaviours kept on purpose (SURVEY 5.9 Q6 and one more found while restating):
  * the epsilon closure is NOT cut below Match and the loop keeps the last accept => longest-on-path, not
    leftmost-first, when the two differ (lazy quantifiers, `a|ab`);
  * FindAllBytes advances `offset += len(Match)` -- the match LENGTH, not its end (compiler.go:646-651) -- so a
    match that does not start at the beginning of the re-sliced input is found again (duplicates).  The reference's
    own test inputs all start with the match.  Reported as quirk Q11 in DESIGN.md; `find_all` reproduces it,
    `find_all_fixed` is the same loop with the offset corrected.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from . import engines as E
from . import syntax as S

Action = Tuple[int, int]  # (tag, offset)


def compact(actions: List[Action]) -> List[Action]:
    last: Dict[int, Action] = {}
    for a in actions:
        last[a[0]] = a
    return sorted(last.values(), key=lambda a: a[0])


class TDFAError(Exception):
    pass


class TDFA:
    def __init__(self, prog: S.Prog, ncap_names: int, max_states: int = 500):
        self.p = prog
        self.ncap_names = ncap_names
        self.max_states = max_states
        self.states: List[List[Tuple[int, List[Action]]]] = []
        self.state_map: Dict[str, int] = {}
        self.trans: List[Dict[int, int]] = []
        self.tag_actions: List[Dict[int, List[Action]]] = []
        self.accept: Dict[int, bool] = {}
        self.accept_eot: Dict[int, bool] = {}
        self.accept_actions: Dict[int, List[Action]] = {}
        self.start_begin = 0
        self.start_any = 0
        self.initial_begin: List[Action] = []
        self.initial_any: List[Action] = []
        self.anchored = E.is_anchored(prog)
        self.build()

    # ---- feasibility (tdfa.go:83-109)
    @staticmethod
    def supported(prog: S.Prog) -> bool:
        for i in prog.inst:
            if i.op == S.InstEmptyWidth and i.arg not in (S.EmptyBeginText, S.EmptyEndText):
                return False
        return True

    def closure(self, states, collect_start: bool, flags: int):
        visited = set()
        result = []
        stack = list(reversed(states))
        while stack:
            sid, acts = stack.pop()
            acts = compact(acts)
            if sid in visited or sid >= len(self.p.inst):
                continue
            visited.add(sid)
            result.append((sid, acts))
            ins = self.p.inst[sid]
            if ins.op == S.InstNop:
                stack.append((ins.out, acts))
            elif ins.op == S.InstCapture:
                na = list(acts)
                if (ins.arg % 2 == 1) or collect_start:
                    na.append((ins.arg, 0))
                stack.append((ins.out, na))
            elif ins.op == S.InstAlt:
                stack.append((ins.arg, acts))
                stack.append((ins.out, acts))
            elif ins.op == S.InstEmptyWidth:
                if (ins.arg & flags) == ins.arg:
                    stack.append((ins.out, acts))
        return result

    @staticmethod
    def key(states) -> str:
        parts = []
        for sid, acts in sorted(states, key=lambda s: s[0]):   # sort.Slice is not stable, but ids are unique in a set
            k = "%d" % sid
            if acts:
                k += "[" + ";".join("%d:%d" % a for a in acts) + "]"
            parts.append(k)
        return ",".join(parts)

    def possible_chars(self, nfa) -> List[int]:
        cs = set()
        for sid, _ in nfa:
            ins = self.p.inst[sid]
            if ins.op == S.InstRune1:
                if ins.rune and ins.rune[0] < 128:
                    cs.add(ins.rune[0])
            elif ins.op == S.InstRune:
                r = ins.rune
                for i in range(0, len(r) - 1, 2):
                    lo, hi = r[i], r[i + 1]
                    if lo < 128:
                        cs.update(range(lo, min(hi, 127) + 1))
            elif ins.op == S.InstRuneAny:
                cs.update(range(128))
            elif ins.op == S.InstRuneAnyNotNL:
                cs.update(c for c in range(128) if c != 10)
        return sorted(cs)

    def transition(self, nfa, c: int):
        nxt = []
        for sid, acts in nfa:
            ins = self.p.inst[sid]
            m = False
            if ins.op == S.InstRune1:
                m = bool(ins.rune) and ins.rune[0] < 128 and ins.rune[0] == c
            elif ins.op == S.InstRune:
                r = ins.rune
                m = any(r[i] <= c <= r[i + 1] for i in range(0, len(r) - 1, 2))
            elif ins.op == S.InstRuneAny:
                m = True
            elif ins.op == S.InstRuneAnyNotNL:
                m = c != 10
            if m:
                nxt.append((ins.out, [(t, o + 1) for t, o in acts]))
        if not nxt:
            return None, None
        res = self.closure(nxt, True, 0)
        if not res:
            return None, None
        common = list(res[0][1])
        for i in range(1, len(res)):
            b = res[i][1]
            k = 0
            while k < len(common) and k < len(b) and common[k] == b[k]:
                k += 1
            common = common[:k]
            if not common:
                break
        if common:
            res = [(sid, acts[len(common):]) for sid, acts in res]
        return res, common

    def _is_accept(self, nfa) -> bool:
        return any(self.p.inst[sid].op == S.InstMatch for sid, _ in nfa)

    def build(self):
        start = [(self.p.start, [])]
        sb = self.closure(start, True, S.EmptyBeginText)
        self.initial_begin = list(sb[0][1]) if sb else []
        self.states = [sb]
        self.state_map[self.key(sb)] = 0
        self.trans = [{}]
        self.tag_actions = [{}]
        if self._is_accept(sb):
            self.accept[0] = True
        sa = self.closure(start, True, 0)
        self.initial_any = list(sa[0][1]) if sa else []
        ka = self.key(sa)
        if ka in self.state_map:
            self.start_any = self.state_map[ka]
        else:
            idx = len(self.states)
            self.states.append(sa)
            self.state_map[ka] = idx
            self.trans.append({})
            self.tag_actions.append({})
            self.start_any = idx
            if self._is_accept(sa):
                self.accept[idx] = True
        work = [0] + ([self.start_any] if self.start_any != 0 else [])
        done = set()
        while work:
            si = work.pop(0)
            if si in done:
                continue
            done.add(si)
            nfa = self.states[si]
            for c in self.possible_chars(nfa):
                nn, actions = self.transition(nfa, c)
                if not nn:
                    continue
                k = self.key(nn)
                ni = self.state_map.get(k)
                if ni is None:
                    ni = len(self.states)
                    if ni >= self.max_states:
                        raise TDFAError("TDFA state explosion: exceeded %d states" % self.max_states)
                    self.states.append(nn)
                    self.state_map[k] = ni
                    self.trans.append({})
                    self.tag_actions.append({})
                    if self._is_accept(nn):
                        self.accept[ni] = True
                    work.append(ni)
                self.trans[si][c] = ni
                if actions:
                    self.tag_actions[si][c] = actions
        for i, nfa in enumerate(self.states):
            for sid, acts in self.closure(nfa, True, S.EmptyEndText):
                if self.p.inst[sid].op == S.InstMatch:
                    self.accept_eot[i] = True
                    if acts:
                        self.accept_actions[i] = compact(acts)
                    break
        for i, nfa in enumerate(self.states):
            if not self.accept.get(i) or i in self.accept_actions:
                continue
            for sid, acts in nfa:
                if self.p.inst[sid].op == S.InstMatch:
                    if acts:
                        self.accept_actions[i] = compact(acts)
                    break

    # ---- the emitted loop (tdfa.go:831-994) and result construction (998-1052)
    def _required_prefix(self) -> Optional[int]:
        return E.Machine(self.p)._required_prefix()

    def find(self, inp: bytes) -> Optional[List[int]]:
        """Raw tags [2*groups] of the first match (-1 = unset), after the result-construction fix-ups."""
        l = len(inp)
        ntags = max(self.ncap_names, 1) * 2
        prefix = self._required_prefix()
        has_prefix = prefix is not None and not self.anchored
        match_end = -1
        match_tags = [-1] * ntags
        start = 0
        while start <= l:
            if has_prefix:
                idx = inp.find(bytes([prefix]), start)
                if idx < 0:
                    break
                start = idx
            tags = [-1] * ntags
            tags[0] = start
            if start == 0:
                state = self.start_begin
                for t, _ in self.initial_begin:
                    tags[t] = start
            else:
                state = self.start_any
                for t, _ in self.initial_any:
                    tags[t] = start
            if self.accept.get(state):
                match_end = start
                match_tags = list(tags)
            if start == l and self.accept_eot.get(state):
                match_end = start
                match_tags = list(tags)
            i = start
            while i < l:
                c = inp[i]
                if c >= 128:
                    break
                ns = self.trans[state].get(c, -1)
                if ns < 0:
                    break
                for t, o in self.tag_actions[state].get(c, []):
                    tags[t] = i + 1 - o
                state = ns
                if self.accept.get(state):
                    for t, o in self.accept_actions.get(state, []):
                        tags[t] = i + 1 - o
                    match_end = i + 1
                    match_tags = list(tags)
                if i == l - 1 and self.accept_eot.get(state):
                    for t, o in self.accept_actions.get(state, []):
                        tags[t] = i + 1 - o
                    match_end = i + 1
                    match_tags = list(tags)
                i += 1
            if match_end >= 0:
                match_tags[1] = match_end
                for g in range(1, self.ncap_names):
                    if match_tags[2 * g] >= 0:
                        if match_tags[2 * g + 1] < 0:
                            match_tags[2 * g + 1] = match_tags[1]
                    else:
                        match_tags[2 * g + 1] = -1      # field left untouched (nil): report the group as unmatched
                return match_tags
            start += 1
        return None

    def find_all(self, inp: bytes, n: int = -1, fixed: bool = False) -> List[List[int]]:
        """compiler.go:602-655 (offset += len(Match): quirk Q11).  Spans are made absolute."""
        res: List[List[int]] = []
        if n == 0:
            return res
        off = 0
        while off < len(inp):
            t = self.find(inp[off:])
            if t is None:
                break
            res.append([x + off if x >= 0 else -1 for x in t])
            if n > 0 and len(res) >= n:
                break
            mlen = t[1] - t[0]
            if mlen > 0:
                off += (t[1] if fixed else mlen)
            else:
                off += 1
        return res

    def find_all_fixed(self, inp: bytes, n: int = -1):
        return self.find_all(inp, n, fixed=True)

    # ---- emitted table view (tdfa.go:584-794) for pinning against the generated files
    def tables(self):
        ns = len(self.states)
        trans = [[self.trans[s].get(c, -1) for c in range(128)] for s in range(ns)]
        cnt = [[len(self.tag_actions[s].get(c, [])) for c in range(128)] for s in range(ns)]
        acts = [[list(map(list, self.tag_actions[s].get(c, []))) for c in range(128)] for s in range(ns)]
        return {"n_states": ns, "transitions": trans, "tag_action_count": cnt, "tag_actions": acts,
                "accept": [bool(self.accept.get(s)) for s in range(ns)],
                "accept_eot": [bool(self.accept_eot.get(s)) for s in range(ns)],
                "accept_actions": [list(map(list, self.accept_actions.get(s, []))) for s in range(ns)]}


def build_for_prog(ast, prog, max_states: int = 500) -> Optional[TDFA]:
    """The reference's decision (compiler.go:137-153, tdfa.go:83-109): the Tagged DFA if it can be built, None if it cannot
    (an empty-width op other than ^/$ of the text, or more than max_states states)."""
    if prog.numcap <= 2 or not TDFA.supported(prog):
        return None
    try:
        t = TDFA(prog, len(S.capture_names(ast)), max_states)
    except TDFAError:
        return None
    if len(t.states) > max_states:
        return None
    return t


def build_for_pattern(pattern: str, max_states: int = 500) -> Optional[TDFA]:
    ast, prog = S.compile_pattern(pattern)
    return build_for_prog(ast, prog, max_states)
