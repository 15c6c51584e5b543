"""Seeded random regular expressions and inputs for the differential tests (host table walk and HIP kernels against
the oracle).  Everything the reference's front end accepts in Perl mode may come out of here: literals, classes, negated
classes, `.`, \\d \\w \\s, capturing / named / non-capturing groups, alternation, greedy and lazy quantifiers, counted
repetition, and (sparingly) the empty-width assertions ^ $ \\b \\B."""
import random

ALPHABET = "ab01-. \n"
ATOMS = ["a", "b", "0", "1", "-", r"\.", r"\d", r"\w", r"\s", ".", "[ab]", "[^a]", "[0-9]", "[a-b0-1]", r"[^\d\n]", "ab", "01", "a-"]
ASSERTS = ["^", "$", r"\b", r"\B", "(?m:^)", "(?m:$)"]


def gen_regex(rng: random.Random, depth: int = 0, allow_assert: bool = True) -> str:
    r = rng.random()
    if depth >= 3 or r < 0.35:
        if allow_assert and rng.random() < 0.06:
            return rng.choice(ASSERTS)
        return rng.choice(ATOMS)
    if r < 0.55:        # concatenation
        return "".join(gen_regex(rng, depth + 1, allow_assert) for _ in range(rng.randrange(2, 4)))
    if r < 0.70:        # alternation inside a group
        alts = "|".join(gen_regex(rng, depth + 1, allow_assert) for _ in range(rng.randrange(2, 4)))
        return rng.choice(["(%s)", "(?:%s)", "(?P<g%d>%%s)" % rng.randrange(100)]) % alts
    if r < 0.92:        # quantifier
        inner = gen_regex(rng, depth + 1, False)
        if len(inner) > 1 and not (inner.startswith("(") and inner.endswith(")")) and not (inner.startswith("[") and inner.endswith("]")) \
                and not (inner.startswith("\\") and len(inner) == 2):
            inner = "(?:%s)" % inner
        q = rng.choice(["*", "+", "?", "{2}", "{1,3}", "{2,}", "{0,2}"])
        if rng.random() < 0.2:
            q += "?"
        return inner + q
    return "(%s)" % gen_regex(rng, depth + 1, allow_assert)


def gen_patterns(seed: int, count: int):
    rng = random.Random(seed)
    out, seen = [], set()
    while len(out) < count:
        p = gen_regex(rng)
        if p in seen or len(p) > 60:
            continue
        seen.add(p)
        out.append(p)
    return out


def gen_input(rng: random.Random, n: int, alphabet: str = ALPHABET) -> bytes:
    """Runs and repeats make matches (and long matches) likely."""
    out = []
    size = 0
    while size < n:
        k = rng.random()
        if k < 0.5:
            s = "".join(rng.choice(alphabet) for _ in range(rng.randrange(1, 8)))
        elif k < 0.8:
            s = rng.choice(alphabet[:6]) * rng.randrange(1, 12)
        else:
            s = rng.choice(["ab", "01", "a-b", "0.1", "ab01", "b-a.0"]) * rng.randrange(1, 5)
        out.append(s)
        size += len(s)
    return "".join(out)[:n].encode()


def has_empty_loop(prog) -> bool:
    """A cycle through instructions that consume nothing (alt / capture / empty-width / nop): the reference's
    backtracker never leaves it unless memoization is on (e.g. `(?:a*)+` -- its Find* functions do not terminate)."""
    from oracle import syntax as S
    n = len(prog.inst)
    succ = [[] for _ in range(n)]
    for i, ins in enumerate(prog.inst):
        if ins.op in (S.InstAlt, S.InstAltMatch):
            succ[i] = [ins.out, ins.arg]
        elif ins.op in (S.InstCapture, S.InstEmptyWidth, S.InstNop):
            succ[i] = [ins.out]
    color = [0] * n
    for s in range(n):
        if color[s]:
            continue
        stack = [(s, 0)]
        color[s] = 1
        while stack:
            v, k = stack[-1]
            if k < len(succ[v]):
                stack[-1] = (v, k + 1)
                w = succ[v][k]
                if color[w] == 1:
                    return True
                if color[w] == 0:
                    color[w] = 1
                    stack.append((w, 0))
            else:
                color[v] = 2
                stack.pop()
    return False


# ---- UTF-8 flavour: multibyte literals and classes over VALID UTF-8 inputs (decoding classes are exact on valid
# UTF-8 only)
ATOMS_U = ["a", "1", "é", "ü", "日", "[α-ω]", r"\p{Greek}", "[^a]", r"[^\d]", "[é日]", r"\pL", r"\PL", ".", r"\w", "aé", "[a-zà-ÿ]"]
ALPHABET_U = ["a", "b", "1", " ", "\n", "é", "ü", "α", "ω", "β", "日", "本", "à", "€", "😀", "-"]


def gen_regex_u(rng: random.Random, depth: int = 0) -> str:
    r = rng.random()
    if depth >= 2 or r < 0.4:
        return rng.choice(ATOMS_U)
    if r < 0.6:
        return "".join(gen_regex_u(rng, depth + 1) for _ in range(rng.randrange(2, 4)))
    if r < 0.75:
        return "(%s)" % "|".join(gen_regex_u(rng, depth + 1) for _ in range(2))
    inner = gen_regex_u(rng, depth + 1)
    if len(inner) > 1 and not (inner[0] in "([" and inner[-1] in ")]") and not (inner[0] == "\\" and len(inner) <= 3):
        inner = "(?:%s)" % inner
    return inner + rng.choice(["*", "+", "?", "{1,2}", "+?"])


def gen_patterns_u(seed: int, count: int):
    rng = random.Random(seed)
    out, seen = [], set()
    while len(out) < count:
        p = gen_regex_u(rng)
        if p not in seen and len(p) <= 40:
            seen.add(p)
            out.append(p)
    return out


def gen_input_u(rng: random.Random, nchars: int) -> bytes:
    return "".join(rng.choice(ALPHABET_U) * rng.choice([1, 1, 1, 2, 5]) for _ in range(nchars)).encode("utf-8")


# ---- texts that EXERCISE a Tagged DFA: random walks through the automaton's own transitions (uniform random text almost never
# gets past the first two bytes of `https?://...`), cut short, restarted and salted with noise so that attempts fail late, accept
# early and go on, overlap, and end exactly at the end of the text (the EOT accepts).
def tdfa_guided_text(tables: dict, rng: random.Random, n: int, noise: float = 0.06) -> bytes:
    trans = tables["transitions"]
    alpha = sorted({c for row in trans for c in range(128) if row[c] >= 0}) or [97]
    out = bytearray()
    st = -1
    while len(out) < n:
        if st < 0:
            st = rng.choice([0, tables.get("start_any", 0)])
        live = [c for c in range(128) if trans[st][c] >= 0]
        r = rng.random()
        if not live or r < noise:
            out.append(rng.choice(alpha) if rng.random() < 0.7 else rng.choice([32, 10, 0xC3, 0xA9, 47, 58]))
            st = -1
            continue
        # prefer bytes that lead to DIFFERENT states (a class of 60 word bytes would otherwise drown the one '.' that moves on)
        by_next = {}
        for c in live:
            by_next.setdefault(trans[st][c], []).append(c)
        c = rng.choice(by_next[rng.choice(sorted(by_next))])
        out.append(c)
        st = trans[st][c]
        if rng.random() < 0.02:
            st = -1
    return bytes(out[:n])


def fuzz_seeds(lo, hi):
    """The seeds of a random-pattern test: its own few by default, RGX_FUZZ_SEEDS=lo:hi for a wide sweep (scripts/gpu_ref_fuzz.sh)."""
    import os
    v = os.environ.get("RGX_FUZZ_SEEDS")
    if v:
        a, b = v.split(":")
        return range(int(a), int(b))
    return range(lo, hi)


def fuzz_default() -> bool:
    """True when a random-pattern test runs its own seeds (its count thresholds apply); False under RGX_FUZZ_SEEDS (a sweep: one seed per
    process, only mismatches count)."""
    import os
    return not os.environ.get("RGX_FUZZ_SEEDS")
