#!/usr/bin/env python3
"""Extract DATA fixtures (inputs + expected outputs) from the reference's own data files:
  data/tactical/tactical_metadata.json     -> tests/golden/tactical_legal_counts.json.gz  (fen, #legal, a legal move)
  data/stockfish_games/**/**.json          -> tests/golden/stockfish_best_moves.json.gz   (fen, best_move = a legal move)
Runs only in the build container (needs /root/reference)."""
import glob, gzip, json, os
REF = os.environ.get("M0_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
d = json.load(open(os.path.join(REF, "data/tactical/tactical_metadata.json")))
rows = [[e["fen"], int(e["legal_moves"]), e.get("move", "")] for e in d]
with gzip.open(os.path.join(OUT, "tactical_legal_counts.json.gz"), "wt") as f:
    json.dump(rows, f, separators=(",", ":"))
out = []
for fn in sorted(glob.glob(os.path.join(REF, "data/stockfish_games/**/*.json"), recursive=True)):
    x = json.load(open(fn))
    pos = x["positions"]
    step = max(1, len(pos) // 500)
    out += [[p["fen"], p["best_move"]] for p in pos[::step][:500]]
with gzip.open(os.path.join(OUT, "stockfish_best_moves.json.gz"), "wt") as f:
    json.dump(out, f, separators=(",", ":"))
print(len(rows), len(out))

# data/eval_games/*.pgn (125 games written by python-chess) -> tests/golden/eval_games_san.json.gz
# one entry per game: [[SAN tokens...], result] -- pins SAN generation (m0_san_*) and whole-game replay
import re
games = []
for fn in sorted(glob.glob(os.path.join(REF, "data/eval_games/*.pgn"))):
    txt = open(fn).read()
    res = re.search(r'\[Result "([^"]+)"\]', txt).group(1)
    body = txt.split("\n\n", 1)[1] if "\n\n" in txt else ""
    body = re.sub(r"\{[^}]*\}", " ", body)
    toks = [t for t in body.split() if not re.fullmatch(r"\d+\.(\.\.)?", t) and t not in ("1-0", "0-1", "1/2-1/2", "*")]
    games.append([toks, res])
with gzip.open(os.path.join(OUT, "eval_games_san.json.gz"), "wt") as f:
    json.dump(games, f, separators=(",", ":"))
print(len(games), sum(len(g[0]) for g in games))
