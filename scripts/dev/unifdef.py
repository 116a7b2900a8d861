#!/usr/bin/env python3
"""scripts/dev/unifdef.py FILE... -D NAME=VALUE ... -U NAME ...

Resolves preprocessor conditionals that depend ONLY on the given macros (value, or undefined with -U), drops the branches
that lose, removes the `#ifndef NAME / #define NAME v / #endif` default blocks and stray `#define NAME` / `#undef NAME` of those
macros, and substitutes remaining uses of valued macros in code by their value.  Conditions that mention any other macro are
left alone.  Used in round 3 to delete the experiment switches whose A/B was lost (DESIGN.md keeps the numbers); the
generated ISA of every translation unit was compared before / after (identical)."""
import re
import sys


def parse_args(argv):
    files, defs, undefs = [], {}, set()
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "-D":
            k, _, v = argv[i + 1].partition("=")
            defs[k] = v if v != "" else "1"
            i += 2
        elif a == "-U":
            undefs.add(argv[i + 1])
            i += 2
        else:
            files.append(a)
            i += 1
    return files, defs, undefs


TOKEN = re.compile(r"\s*(defined\s*\(\s*\w+\s*\)|defined\s+\w+|\w+|&&|\|\||==|!=|>=|<=|[!()<>])")


def evaluate(expr, defs, undefs):
    """value of a #if expression, or None if it involves a macro we do not know"""
    expr = re.sub(r"//.*$", "", expr).strip()
    expr = re.sub(r"/\*.*?\*/", "", expr).strip()
    toks, pos = [], 0
    while pos < len(expr):
        m = TOKEN.match(expr, pos)
        if not m:
            return None
        toks.append(m.group(1))
        pos = m.end()
    out = []
    for t in toks:
        m = re.match(r"defined\s*\(?\s*(\w+)\s*\)?", t)
        if m:
            n = m.group(1)
            if n in defs:
                out.append("1")
            elif n in undefs:
                out.append("0")
            else:
                return None
        elif re.match(r"^\d+$", t):
            out.append(t)
        elif re.match(r"^\w+$", t):
            if t in defs:
                v = defs[t]
                if not re.match(r"^-?\d+$", v):
                    return None
                out.append(v)
            elif t in undefs:
                out.append("0")
            else:
                return None
        else:
            out.append({"&&": " and ", "||": " or ", "!": " not "}.get(t, t))
    try:
        return bool(eval(" ".join(out)))
    except Exception:
        return None


def partial(line, defs, undefs):
    """a conditional that also depends on other macros: the known ones are substituted in its text"""
    def rep_defined(m):
        n = m.group(1)
        return "1" if n in defs else ("0" if n in undefs else m.group(0))
    line = re.sub(r"defined\s*\(\s*(\w+)\s*\)", rep_defined, line)
    for k, v in defs.items():
        line = re.sub(r"\b%s\b" % re.escape(k), v, line)
    for k in undefs:
        line = re.sub(r"\b%s\b" % re.escape(k), "0", line)
    return line


def process(text, defs, undefs):
    lines = text.split("\n")
    out = []
    # stack entries: dict(kind: 'known'|'unknown', taken: bool (a branch already taken), active: bool (current branch emitted))
    stack = []
    known = set(defs) | undefs

    def emitting():
        return all(f["active"] for f in stack)

    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if m:
            d, rest = m.group(1), m.group(2).strip()
            if d in ("if", "ifdef", "ifndef"):
                if d == "ifdef":
                    cond = "defined(%s)" % rest.split()[0]
                elif d == "ifndef":
                    cond = "!defined(%s)" % rest.split()[0]
                else:
                    cond = rest
                # default block  #ifndef X / #define X v / #endif  of a known macro
                if d == "ifndef" and rest.split()[0] in known and i + 2 < len(lines) and re.match(r"#\s*define\s+%s\b" % rest.split()[0], lines[i + 1].strip()) and re.match(r"#\s*endif", lines[i + 2].strip()):
                    if emitting():
                        pass  # dropped
                    i += 3
                    continue
                val = evaluate(cond, defs, undefs) if emitting() else False
                if not emitting():
                    stack.append({"kind": "dead", "taken": True, "active": False})
                elif val is None:
                    stack.append({"kind": "unknown", "taken": False, "active": True})
                    out.append(partial(ln, defs, undefs) if d == "if" else ln)
                else:
                    stack.append({"kind": "known", "taken": val, "active": val})
            elif d == "elif":
                f = stack[-1]
                if f["kind"] == "dead":
                    pass
                elif f["kind"] == "unknown":
                    out.append(partial(ln, defs, undefs))
                else:
                    if f["taken"]:
                        f["active"] = False
                    else:
                        val = evaluate(rest, defs, undefs)
                        if val is None:
                            raise SystemExit(f"cannot resolve '#elif {rest}' after a resolved #if")
                        f["active"] = val
                        f["taken"] = val
            elif d == "else":
                f = stack[-1]
                if f["kind"] == "dead":
                    pass
                elif f["kind"] == "unknown":
                    out.append(ln)
                else:
                    f["active"] = not f["taken"]
                    f["taken"] = True
            else:  # endif
                f = stack.pop()
                if f["kind"] == "unknown":
                    out.append(ln)
            i += 1
            continue
        if emitting():
            dm = re.match(r"#\s*(define|undef)\s+(\w+)\b", s)
            if dm and dm.group(2) in known:
                i += 1
                continue  # stray definition of a resolved macro
            if not s.startswith("#"):
                for k, v in defs.items():
                    ln = re.sub(r"\b%s\b" % re.escape(k), v, ln)
            out.append(ln)
        i += 1
    if stack:
        raise SystemExit("unbalanced conditionals")
    return "\n".join(out)


if __name__ == "__main__":
    files, defs, undefs = parse_args(sys.argv[1:])
    for f in files:
        src = open(f).read()
        dst = process(src, defs, undefs)
        if dst != src:
            open(f, "w").write(dst)
            print("rewrote", f)
