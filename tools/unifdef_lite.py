"""A small unifdef: resolves #if / #ifdef / #ifndef / #elif / #else / #endif blocks whose conditions only name macros given on the command line (NAME=VALUE), drops the
`#ifndef NAME / #define NAME v / #endif` that introduced them, and leaves everything else alone.  Round 5: the measured-negative compile-time arms of the pipelined decode
kernels leave the product sources; the arms and their results stay under profiles/negatives/.   usage: python tools/unifdef_lite.py FILE NAME=VALUE ..."""
import re, sys

def main():
    path = sys.argv[1]
    known = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        known[k] = int(v)
    ident = re.compile(r"[A-Za-z_][A-Za-z0-9_]*")

    def evaluate(expr):
        """value of a preprocessor expression, or None if it names anything unknown"""
        e = re.sub(r"//.*$", "", expr).strip()
        e = re.sub(r"defined\s*\(\s*([A-Za-z_][A-Za-z0-9_]*)\s*\)", lambda m: "1" if m.group(1) in known else "__UNKNOWN__", e)
        for name in set(ident.findall(e)):
            if name in known:
                e = re.sub(r"\b%s\b" % name, str(known[name]), e)
            else:
                return None
        e = e.replace("&&", " and ").replace("||", " or ")
        e = re.sub(r"!(?!=)", " not ", e)
        try:
            return int(bool(eval(e, {"__builtins__": {}}, {})))
        except Exception:
            return None

    lines = open(path).read().split("\n")
    out = []
    # stack entries: dict(mode='resolved'|'keep', taken=bool (resolved: has a branch been taken), emitting=bool)
    stack = []
    def emitting():
        return all(s["emit"] for s in stack)
    i = 0
    while i < len(lines):
        ln = lines[i]
        st = ln.strip()
        m = re.match(r"#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)", st)
        if not m:
            if emitting():
                out.append(ln)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2)
        if kw in ("if", "ifdef", "ifndef"):
            # the knob's own definition: #ifndef NAME / #define NAME v / #endif  -> dropped
            if kw == "ifndef" and rest.strip().split()[0] in known and i + 2 < len(lines) and re.match(r"#\s*define\s+%s\b" % rest.strip().split()[0], lines[i + 1].strip()) and lines[i + 2].strip().startswith("#endif"):
                i += 3
                continue
            if kw == "if":
                v = evaluate(rest)
            else:
                name = rest.strip().split()[0]
                v = None if name not in known else (1 if kw == "ifdef" else 0)
            if v is None:
                stack.append({"mode": "keep", "emit": True})
                if emitting():
                    out.append(ln)
            else:
                stack.append({"mode": "resolved", "taken": bool(v), "emit": bool(v)})
        elif kw == "elif":
            s = stack[-1]
            if s["mode"] == "keep":
                if emitting():
                    out.append(ln)
            else:
                if s["taken"]:
                    s["emit"] = False
                else:
                    v = evaluate(rest)
                    if v is None:
                        raise SystemExit("%s:%d: #elif with unknown macros after a resolved #if" % (path, i + 1))
                    s["emit"] = bool(v); s["taken"] = bool(v)
        elif kw == "else":
            s = stack[-1]
            if s["mode"] == "keep":
                if emitting():
                    out.append(ln)
            else:
                s["emit"] = not s["taken"]; s["taken"] = True
        else:
            s = stack.pop()
            if s["mode"] == "keep" and emitting():
                out.append(ln)
        i += 1
    assert not stack
    open(path, "w").write("\n".join(out))

main()
