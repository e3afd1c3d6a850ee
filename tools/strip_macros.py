#!/usr/bin/env python3
"""Partial preprocessor for the kernel skeletons (round 5, freezing the product surface): resolves the #if / #ifdef /
#ifndef / #elif directives whose condition mentions ONLY macros given on the command line, drops the dead branches and
the `#ifndef X / #define X v / #endif` default blocks of those macros, and replaces the macros' remaining uses by their
values.  Everything else is left as it stands.
usage: strip_macros.py FILE NAME=VALUE [NAME=VALUE ...]   (rewrites FILE in place; --check only reports)"""
import re
import sys

IDENT = re.compile(r"\b[A-Za-z_]\w*\b")


def evaluate(expr, known):
    """value of a preprocessor expression if every identifier in it is known, else None"""
    expr = re.sub(r"//.*$", "", expr).strip()
    expr = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", lambda m: "1" if (m.group(1) or m.group(2)) in known else "@", expr)
    if "@" in expr:
        return None
    names = set(IDENT.findall(expr))
    if not names <= set(known):
        return None
    py = IDENT.sub(lambda m: "(%s)" % known[m.group(0)], expr)
    py = py.replace("&&", " and ").replace("||", " or ")
    py = re.sub(r"!(?!=)", " not ", py)
    try:
        return bool(eval(py, {"__builtins__": {}}, {}))  # noqa: S307 (our own headers)
    except Exception:  # noqa: BLE001
        return None


def strip(text, known):
    lines = text.split("\n")
    out = []
    # stack of frames: dict(kind='resolved'|'kept', taken=bool (a branch was already chosen), live=bool)
    stack = []

    def live():
        return all(f["live"] for f in stack)

    i = 0
    while i < len(lines):
        line = lines[i]
        s = line.strip()
        m = re.match(r"#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", s)
        if not m:
            if live():
                out.append(line)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2).strip()
        if kw in ("if", "ifdef", "ifndef"):
            if kw == "if":
                val = evaluate(rest, known)
            else:
                name = rest.split()[0]
                val = None
                if name in known:
                    val = (kw == "ifdef")
            # the default block of a stripped macro:  #ifndef X / #define X ... / #endif
            if kw == "ifndef" and rest.split()[0] in known:
                j = i + 1
                while j < len(lines) and not lines[j].strip().startswith("#endif"):
                    j += 1
                body = [b.strip() for b in lines[i + 1:j] if b.strip() and not b.strip().startswith("//")]
                if all(b.startswith("#define " + rest.split()[0]) for b in body):
                    i = j + 1
                    continue
            if val is None:
                stack.append({"kind": "kept", "live": True, "taken": False})
                if live():
                    out.append(line)
            else:
                stack.append({"kind": "resolved", "live": val, "taken": val})
        elif kw == "elif":
            f = stack[-1]
            if f["kind"] == "kept":
                val = evaluate(rest, known)
                if f.get("closed"):
                    f["live"] = False  # (behind a branch that is always taken)
                elif val is None:
                    f["live"] = True
                    if all(g["live"] for g in stack):
                        out.append(line)
                elif val:
                    f["live"], f["closed"] = True, True
                    if all(g["live"] for g in stack):
                        out.append(re.sub(r"#(\s*)elif.*", r"#\1else", line, count=1))
                else:
                    f["live"] = False  # a branch never taken
            else:
                if f["taken"]:
                    f["live"] = False
                else:
                    val = evaluate(rest, known)
                    if val is None:
                        # becomes the head of a kept conditional
                        f["kind"], f["live"] = "kept", True
                        if live():
                            out.append(re.sub(r"#(\s*)elif", r"#\1if", line, count=1))
                    else:
                        f["live"], f["taken"] = val, val
        elif kw == "else":
            f = stack[-1]
            if f["kind"] == "kept":
                if f.get("closed"):
                    f["live"] = False
                else:
                    f["live"] = True
                    if live():
                        out.append(line)
            else:
                f["live"] = not f["taken"]
                f["taken"] = True
        else:  # endif
            f = stack.pop()
            if f["kind"] == "kept" and live():
                out.append(line)
        i += 1
    text = "\n".join(out)
    # remaining uses (C++ level, not comments): the value itself
    for name, value in known.items():
        text = re.sub(r"^\s*#\s*define\s+%s\b.*\n" % name, "", text, flags=re.M)
    lines = []
    for line in text.split("\n"):
        code, sep, comment = line.partition("//")
        for name, value in known.items():
            code = re.sub(r"\b%s\b" % name, value if re.fullmatch(r"-?\w+", value) else "(%s)" % value, code)
        m = re.match(r"(\s*#\s*(?:if|elif)\s+)(.*)", code)
        if m:
            code = m.group(1) + simplify(m.group(2))
        lines.append(code + sep + comment)
    return "\n".join(lines)


def simplify(expr):
    """drops constant clauses of a top-level conjunction / disjunction: `4 == 4 && X` -> `X`"""
    for op, neutral in (("&&", True), ("||", False)):
        if op in expr and ("&&" if op == "||" else "||") not in expr and "(" not in expr.replace("defined(", ""):
            kept = []
            for clause in (c.strip() for c in expr.split(op)):
                v = evaluate(clause, {})
                if v is None:
                    kept.append(clause)
                elif v != neutral:
                    return "0" if op == "&&" else "1"
            return (" %s " % op).join(kept) if kept else ("1" if neutral else "0")
    return expr


def main():
    check = "--check" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--check"]
    path, known = args[0], dict(a.split("=", 1) for a in args[1:])
    old = open(path).read()
    new = strip(old, known)
    print("%s: %d -> %d lines" % (path, old.count("\n"), new.count("\n")))
    if not check:
        open(path, "w").write(new)


if __name__ == "__main__":
    main()
