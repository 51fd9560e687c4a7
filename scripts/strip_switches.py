#!/usr/bin/env python3
"""Resolves preprocessor conditionals on a given set of macros and leaves every other line alone (a partial `unifdef`): the tool that retired
the A/B and ablation switches from rustfhe_amd/csrc/ in round 5.  A conditional is resolved only when every identifier in it is in the set.

    strip_switches.py file.hpp NAME=value ... NAME= (defined, empty) ... -NAME (undefined)   # rewrites the file in place

`#ifndef NAME / #define NAME v / #endif` knob blocks disappear when NAME is given a value: the caller then writes the constant where it is used.
"""
import re
import sys


def evaluate(expr, macros):
    """None if the expression mentions an identifier we do not know, else its truth value."""
    e = expr.split("//")[0].strip()
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: ("@%s@" % m.group(1)), e)
    e = re.sub(r"defined\s+(\w+)", lambda m: ("@%s@" % m.group(1)), e)
    for name in re.findall(r"@(\w+)@", e):
        if name not in macros:
            return None
    e = re.sub(r"@(\w+)@", lambda m: "1" if macros[m.group(1)] is not None else "0", e)
    for name in set(re.findall(r"[A-Za-z_]\w*", e)):
        if name not in macros:
            return None
        v = macros[name]
        e = re.sub(r"\b%s\b" % name, "0" if v in (None, "") else "(%s)" % v, e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    return bool(eval(e, {"__builtins__": {}}))


def strip(lines, macros):
    out = []
    # stack entries: [resolved?, emitting-this-branch, some-branch-already-taken, parent_emitting]
    stack = []
    emitting = lambda: all(s[1] for s in stack if s[0]) if stack else True
    for line in lines:
        s = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting():
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ("ifdef", "ifndef", "if"):
            if kind == "if":
                val = evaluate(rest, macros)
            else:
                name = rest.split()[0]
                val = None if name not in macros else ((macros[name] is not None) == (kind == "ifdef"))
            if val is None:
                if emitting():
                    out.append(line)
                stack.append([False, True, False])
            else:
                stack.append([True, val, val])
        elif kind == "elif":
            top = stack[-1]
            if not top[0]:
                if emitting():
                    out.append(line)
            else:
                val = evaluate(rest, macros)
                if val is None:
                    raise SystemExit("unresolvable #elif after a resolved #if: " + line)
                top[1] = (not top[2]) and val
                top[2] = top[2] or val
        elif kind == "else":
            top = stack[-1]
            if not top[0]:
                if emitting():
                    out.append(line)
            else:
                top[1] = not top[2]
                top[2] = True
        else:
            top = stack.pop()
            if not top[0] and emitting():
                out.append(line)
    assert not stack
    return out


if __name__ == "__main__":
    path, macros = sys.argv[1], {}
    for a in sys.argv[2:]:
        if a.startswith("-"):
            macros[a[1:]] = None
        else:
            k, _, v = a.partition("=")
            macros[k] = v
    src = open(path).read().split("\n")
    res = strip(src, macros)
    open(path, "w").write("\n".join(res))
    print("%s: %d -> %d lines" % (path, len(src), len(res)))
