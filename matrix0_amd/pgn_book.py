"""Opening book from a PGN file: `load_opening_book` / `get_opening_position` of the reference worker
(azchess/selfplay/internal.py:39-69), which reads the games with python-chess (chess.pgn.read_game) and keeps a copy of the
board after each of the first 20 plies of every game, up to 50 000 positions; a new self-play game starts from
random.choice(OPENING_BOOK).

Here the movetext is parsed directly: every SAN token is matched against the SAN of the legal moves of the current position
(m0_san_legal_fen: python-chess Board.san() semantics, pinned by the 7 875 SAN tokens of the reference's own PGN files,
tests/test_san_pgn.py), the position is advanced with m0_fen_after, and the book is handed to the engine as FEN strings
(m0_selfplay_set_openings), which picks with the game's own stream as random.choice would.  Host code, no GPU.

Known deviation: the reference stores board.copy() WITH its move stack, so the book line's moves count towards repetition /
claim_draw in the game that starts there; a FEN carries no history, so a game started from the book begins with an empty
repetition window (a repetition that needs pre-book positions is seen 1-2 occurrences later than the reference would)."""
from __future__ import annotations

import re
from pathlib import Path
from typing import List, Optional

from . import engine as eng

START_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
RESULTS = {"1-0", "0-1", "1/2-1/2", "*"}
_TAG = re.compile(r'^\s*\[(\w+)\s+"((?:[^"\\]|\\.)*)"\]\s*$')
_TOKEN = re.compile(r'\{[^}]*\}|\(|\)|\$\d+|\d+\.(?:\.\.)?|\.\.\.|[^\s(){}]+')


def _norm(tok: str) -> str:
    tok = tok.rstrip("!?")
    if tok.startswith(("0-0-0", "0-0")):
        tok = tok.replace("0", "O")
    tok = tok.rstrip("+#")
    # python-chess's SAN reader also takes a promotion written without '=' ("e8Q", "dxe8N")
    m = re.fullmatch(r"(.*[a-h][18])([QRBN])", tok)
    if m and "=" not in tok:
        tok = m.group(1) + "=" + m.group(2)
    return tok


def _games(text: str):
    """Yield (headers, movetext) per game: a tag section, then movetext up to the next tag section."""
    headers, moves = {}, []
    for line in text.splitlines():
        if line.startswith("%"):                       # PGN escape line
            continue
        m = _TAG.match(line)
        if m:
            if "".join(moves).strip():                 # a tag pair after movetext starts the next game
                yield headers, "\n".join(moves)
                headers, moves = {}, []
            headers[m.group(1)] = m.group(2)
            continue
        moves.append(line.split(";", 1)[0])            # ';' comments run to the end of the line
    if headers or "".join(moves).strip():
        yield headers, "\n".join(moves)


def mainline_fens(headers: dict, movetext: str, max_plies: int) -> List[str]:
    """FEN after each of the first `max_plies` mainline moves (variations, comments, NAGs and move numbers skipped; parsing of a
    game stops at its first token that is not the SAN of a legal move, as python-chess records an error and ends the line)."""
    # python-chess's Game.board() starts from the FEN header whenever there is one (SetUp is not consulted)
    fen = headers.get("FEN") or START_FEN
    out: List[str] = []
    depth = 0
    for tok in _TOKEN.findall(movetext):
        if tok.startswith("{") or tok.startswith("$"):
            continue
        if tok == "(":
            depth += 1
            continue
        if tok == ")":
            depth = max(0, depth - 1)
            continue
        if depth > 0:
            continue
        if tok in RESULTS:
            break
        if re.fullmatch(r"\d+\.(?:\.\.)?|\.\.\.", tok):
            continue
        tok = re.sub(r"^\d+\.(?:\.\.)?", "", tok)    # "1.e4" written without a space
        if not tok:
            continue
        if len(out) >= max_plies:
            break
        want = _norm(tok)
        hit: Optional[str] = None
        for uci, san in eng.san_legal(fen):
            if _norm(san) == want:
                hit = uci
                break
        if hit is None:
            break
        fen = eng.fen_after(fen, [hit])
        out.append(fen)
    return out


def load_opening_book(pgn_path: str, max_positions: int = 50000, plies_per_game: int = 20) -> List[str]:
    """Positions of the book as FEN strings, in file order (internal.py:39-63); [] when the file does not exist."""
    p = Path(pgn_path)
    if not p.exists():
        return []
    book: List[str] = []
    for headers, movetext in _games(p.read_text(errors="replace")):
        if len(book) >= max_positions:                 # the reference tests the cap once per game (internal.py:49)
            break
        book.extend(mainline_fens(headers, movetext, plies_per_game))
    return book
