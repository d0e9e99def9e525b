"""Version metadata of the `sdod` package (same public names as the reference's sdod/version.py:1-47:
`version`, `repo`, `commit`, `has_repo`, `info()`); repository probing is optional and never fatal."""
import os
import subprocess

version = '0.1.0.dev0'
repo = 'unknown'
commit = 'unknown'
has_repo = False


def _git(*args):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run(('git', '-C', root) + args, capture_output=True, text=True, timeout=5)
    if out.returncode != 0:
        raise RuntimeError(out.stderr)
    return out.stdout.strip()


def _probe():
    global repo, commit, has_repo
    try:
        head = _git('rev-parse', 'HEAD')
    except Exception:
        return
    has_repo = True
    commit = head
    try:
        repo = _git('remote', 'get-url', 'origin')
    except Exception:
        repo = 'local'
    try:
        flags = []
        if _git('status', '--porcelain', '--untracked-files=no'):
            flags.append('dirty')
        if flags:
            commit += ' ({})'.format(','.join(flags))
    except Exception:
        pass


_probe()

__all__ = ['version', 'repo', 'commit', 'has_repo']


def info():
    return {name: globals()[name] for name in __all__}
