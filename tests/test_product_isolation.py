"""The oracle is test infrastructure: nothing under redisearch_amd/ (Python or C++) may import, include, link or
call it, and the product library must not carry a CPU fallback entry point."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_package_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "redisearch_amd")):
        if "lib" in base.split(os.sep) or "__pycache__" in base:
            continue
        for f in files:
            if not f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                continue
            text = open(os.path.join(base, f), encoding="utf-8", errors="replace").read()
            if re.search(r"^\s*(import|from)\s+oracle\b", text, re.M) or re.search(r'#include\s*[<"].*oracle', text) \
                    or "liboracle" in text:
                bad.append(os.path.relpath(os.path.join(base, f), ROOT))
    assert not bad, bad


def test_allowed_oracle_users_only():
    users = []
    for f in os.listdir(ROOT):
        if f.endswith(".py"):
            if re.search(r"^\s*(import|from)\s+oracle\b", open(os.path.join(ROOT, f)).read(), re.M):
                users.append(f)
    assert sorted(users) == ["__graft_entry__.py", "bench.py"], users


def test_scripts_do_not_use_the_oracle():
    # measurement scripts that need the oracle (as encoder / checker / CPU baseline) live under tests/
    users = []
    d = os.path.join(ROOT, "scripts")
    for f in os.listdir(d):
        if f.endswith((".py", ".sh")) and re.search(r"(^\s*(import|from)\s+oracle\b|liboracle)", open(os.path.join(d, f)).read(), re.M):
            users.append(f)
    assert not users, users
