"""Development check (this container only: it reads /root/reference): identifier-blind token similarity of every function of the product
and the tests against every function of the reference's hot-path files.  Names become N, strings S, numbers stay, keywords and operators
stay; difflib ratio over the token streams of functions of >= 120 tokens.  Prints pairs above the threshold (default 0.6)."""
import ast
import difflib
import io
import keyword
import sys
import tokenize
import glob

REF = ['/root/reference/modules/uberBlast.py', '/root/reference/modules/clust.py', '/root/reference/modules/configure.py', '/root/reference/PEPPAN.py']


def functions(path):
    src = open(path).read()
    try:
        tree = ast.parse(src)
    except SyntaxError:
        return []
    lines = src.splitlines(True)
    out = []
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            text = ''.join(lines[node.lineno - 1:node.end_lineno])
            out.append((node.name, node.lineno, text))
    return out


def stream(text):
    import textwrap
    toks = []
    try:
        for t in tokenize.generate_tokens(io.StringIO(textwrap.dedent(text)).readline):
            if t.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENDMARKER):
                continue
            if t.type == tokenize.NAME:
                toks.append(t.string if keyword.iskeyword(t.string) else 'N')
            elif t.type == tokenize.STRING:
                if toks and toks[-1] in (':', '(') or not toks:
                    pass
                toks.append('S')
            else:
                toks.append(t.string)
    except (tokenize.TokenError, IndentationError):
        pass
    return toks


def main():
    thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
    ref = []
    for p in REF:
        for name, line, text in functions(p):
            s = stream(text)
            if len(s) >= 60:
                ref.append((p, name, line, s))
    mine = sorted(glob.glob('peppan_amd/*.py') + glob.glob('tests/*.py') + glob.glob('oracle/*.py') + ['bench.py'])
    hits = []
    for p in mine:
        for name, line, text in functions(p):
            s = stream(text)
            if len(s) < 120:
                continue
            for rp, rname, rline, rs in ref:
                m = difflib.SequenceMatcher(None, s, rs, autojunk=False)
                if m.real_quick_ratio() < thr or m.quick_ratio() < thr:
                    continue
                r = m.ratio()
                if r >= thr:
                    hits.append((r, '%s:%d %s' % (p, line, name), '%s:%d %s' % (rp, rline, rname)))
    for r, a, b in sorted(hits, reverse=True):
        print('%.2f  %-60s %s' % (r, a, b))
    print('%d pairs at or above %.2f' % (len(hits), thr))


if __name__ == '__main__':
    main()
