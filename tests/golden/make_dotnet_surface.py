#!/usr/bin/env python3
"""Extract the member surface of the four reference files the C# shim replaces into
tests/golden/dotnet_surface.json: every public / protected / internal member of every type those files declare
(kind, name, number of parameters), plus every `Type.Member` the rest of the reference's sources names statically
(that is how `Deflater.CompressionLevel` shows up: Zip/FastZip.cs:342).  The json holds names and arities only - data, no source text.

    python tests/golden/make_dotnet_surface.py            # needs /root/reference; rewrites the json

tests/test_dotnet_surface.py checks sharpziplib_amd/dotnet/*.cs against the json with the same extractor (and, where
/root/reference exists, that the json is current)."""
import json
import os
import re
import sys

REF_SRC = "/root/reference/src/ICSharpCode.SharpZipLib"
REPLACED = ["Zip/Compression/Deflater.cs", "Zip/Compression/Inflater.cs",
            "Zip/Compression/Streams/InflaterInputStream.cs", "Zip/Compression/Streams/DeflaterOutputStream.cs"]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dotnet_surface.json")

MODS = {"public", "protected", "internal", "private", "static", "virtual", "override", "abstract", "sealed", "unsafe",
        "readonly", "const", "new", "extern", "async", "partial", "volatile"}
VISIBLE = {"public", "protected", "internal"}


def strip(text):
    """Comments, string and char literals, preprocessor lines and attributes' contents out; layout kept."""
    out = []
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in text[i:j]))
            i = j
        elif c == '"' or (c in "@$" and i + 1 < n and text[i + 1] == '"'):
            verbatim = c == "@"
            if c != '"':
                i += 1
            i += 1
            while i < n:
                if verbatim and text.startswith('""', i):
                    i += 2
                elif not verbatim and text[i] == "\\":
                    i += 2
                elif text[i] == '"':
                    i += 1
                    break
                else:
                    i += 1
            out.append('""')
        elif c == "'":
            j = i + 1
            while j < n and text[j] != "'":
                j += 2 if text[j] == "\\" else 1
            out.append("' '")
            i = j + 1
        else:
            out.append(c)
            i += 1
    lines = ["" if ln.lstrip().startswith("#") else ln for ln in "".join(out).split("\n")]
    return "\n".join(lines)


def count_params(plist):
    plist = plist.strip()
    if not plist:
        return 0
    depth, k = 0, 1
    for ch in plist:
        if ch in "<([":
            depth += 1
        elif ch in ">)]":
            depth -= 1
        elif ch == "," and depth == 0:
            k += 1
    return k


def parse_header(h, type_name, in_enum):
    """h: the text of one member declaration up to its body.  -> list of (visibility-set, kind, name, arity)."""
    h = re.sub(r"\[[^\[\]]*\]", " ", h)              # attributes
    h = " ".join(h.split())
    if not h:
        return []
    if in_enum:
        return [({"public"}, "enumvalue", m, -1) for m in re.findall(r"(?:^|,)\s*(\w+)", re.sub(r"=[^,]*", "", h))]
    toks = h.split(" ")
    mods = set()
    while toks and toks[0] in MODS:
        mods.add(toks.pop(0))
    rest = " ".join(toks)
    m = re.match(r"(class|struct|enum|interface)\s+(\w+)", rest)
    if m:
        return [(mods, "type", m.group(2), -1)]
    if "const" in mods or "(" not in rest.split("=")[0]:
        # field / constant list: `int A = 9, B = 1`  (properties arrive here with their `{` cut off: told apart by the caller)
        decl = rest
        while re.search(r"\([^()]*\)", decl):
            decl = re.sub(r"\([^()]*\)", "", decl)
        first = decl.split("=")[0].split()
        names = []
        if len(first) >= 2:
            names.append(first[-1])
            for part in re.split(r",(?![^<]*>)", decl)[1:]:
                nm = part.split("=")[0].strip().split()
                if len(nm) == 1:
                    names.append(nm[0])
        return [(mods, "field", nm, -1) for nm in names]
    k = rest.find("(")
    if k > 0:
        m = re.search(r"(~?\w+)\s*$", rest[:k])
        d, j = 0, k
        while j < len(rest):
            d += rest[j] == "("
            d -= rest[j] == ")"
            if d == 0:
                break
            j += 1
        if m and j < len(rest):
            name = m.group(1)
            kind = "ctor" if name == type_name else "method"
            return [(mods, kind, name, count_params(rest[k + 1:j]))]
    m = re.match(r"[\w<>\[\],\.\?\* ]+?\s+this\s*\[", rest)
    if m:
        return [(mods, "indexer", "this", -1)]
    return []


def members(text):
    """{type name: {(kind, name, arity), ...}} for the visible members of every type declared in text."""
    t = strip(text)
    res = {}
    stack = []          # (type name or None, is_enum)
    i, n = 0, len(t)
    start = 0           # start of the current declaration header
    paren = 0
    while i < n:
        c = t[i]
        if c == "(":
            paren += 1
        elif c == ")":
            paren -= 1
        elif paren == 0 and (c == "{" or c == ";" or t.startswith("=>", i) or (c == "}" )):
            header = t[start:i]
            cur = stack[-1] if stack else (None, False, False)
            in_type = bool(stack) and cur[0] is not None and cur[2]
            if c == "}":
                if in_type and cur[1] and header.strip():
                    for mods, kind, name, ar in parse_header(header, cur[0], True):
                        res[cur[0]].add((kind, name, ar))
                if stack:
                    stack.pop()
                i += 1
                start = i
                continue
            decls = []
            hs = " ".join(re.sub(r"\[[^\[\]]*\]", " ", header).split())
            is_ns = hs.startswith("namespace ")
            if in_type and not cur[1]:
                decls = parse_header(header, cur[0], False)
            elif (not stack or stack[-1][0] is None or is_ns or not stack[-1][2]) and not is_ns:
                decls = [d for d in parse_header(header, "", False) if d[1] == "type"]
            if in_type and not cur[1] and c == "{" and decls and decls[0][1] == "field":
                decls = [(decls[0][0], "property", decls[0][2], -1)]           # `T Name {` is a property
            if in_type and not cur[1] and t.startswith("=>", i) and decls and decls[0][1] == "field":
                decls = [(decls[0][0], "property", decls[0][2], -1)]
            for mods, kind, name, ar in decls:
                if kind == "type":
                    res.setdefault(name, set())
                if in_type and (mods & VISIBLE):
                    res[cur[0]].add((kind, name, ar))
            if c == "{":
                if is_ns:
                    stack.append((None, False, False))
                elif decls and decls[0][1] == "type":
                    is_enum = bool(re.search(r"\benum\s+" + decls[0][2], hs))
                    stack.append((decls[0][2], is_enum, True))
                else:
                    stack.append((cur[0], False, False))       # a method / accessor body: skip its statements
                i += 1
                start = i
                continue
            if t.startswith("=>", i):
                # expression body: skip to the terminating ';' (or to the accessor list's '}')
                j = i + 2
                d = 0
                while j < n and not (t[j] == ";" and d == 0):
                    if t[j] in "({":
                        d += 1
                    elif t[j] in ")}":
                        d -= 1
                    j += 1
                i = j + 1
                start = i
                continue
            i += 1
            start = i
            continue
        i += 1
    return res


def static_uses(root, type_names, skip):
    """{type: {member}}: `Type.Member` named anywhere in the sources outside the replaced files."""
    uses = {}
    pat = re.compile(r"(?<![\w\.])(" + "|".join(sorted(type_names)) + r")\.(\w+)")
    for d, _, files in os.walk(root):
        for f in sorted(files):
            if not f.endswith(".cs"):
                continue
            p = os.path.join(d, f)
            rel = os.path.relpath(p, root)
            if rel in skip:
                continue
            for ty, mem in pat.findall(strip(open(p, encoding="utf-8-sig").read())):
                uses.setdefault(ty, {}).setdefault(mem, rel)
    return uses


def surface_of_reference():
    types = {}
    for rel in REPLACED:
        for ty, mem in members(open(os.path.join(REF_SRC, rel), encoding="utf-8-sig").read()).items():
            types.setdefault(ty, set()).update(mem)
    uses = static_uses(REF_SRC, set(types), set(REPLACED))
    # a `Type.X` use counts when X is not a namespace-style qualifier of something else: keep those that name a member
    # of the reference's own type (everything else is e.g. a local variable called like the type)
    used = {}
    for ty, mm in uses.items():
        names = {m[1] for m in types[ty]}
        for mem, where in mm.items():
            if mem in names:
                used.setdefault(ty, {})[mem] = where
    return {"files": REPLACED,
            "types": {ty: sorted([list(m) for m in mem]) for ty, mem in sorted(types.items())},
            "static_uses": {ty: dict(sorted(m.items())) for ty, m in sorted(used.items())}}


if __name__ == "__main__":
    if not os.path.isdir(REF_SRC):
        sys.exit("needs " + REF_SRC)
    s = surface_of_reference()
    json.dump(s, open(OUT, "w"), indent=1, sort_keys=True)
    print("%s: %d types, %d members" % (OUT, len(s["types"]), sum(len(v) for v in s["types"].values())))
