"""Build dictionary-kernel variants with different temporal hints: scratch/_libs/hint_<own><gather><prev><store>.so"""
import sys, os
sys.path.insert(0, ".")
from concurrent.futures import ThreadPoolExecutor
from bodge_amd.build import build_library
variants = sys.argv[1:] or ["0000", "1010", "1110", "0011", "0010", "1000", "1011", "0001"]
def one(v):
    out = os.path.abspath(f"scratch/_libs/hint_{v}.so")
    build_library(force=True, verbose=False, output=out,
                  defines=(f"BDG_HINT_OWN={v[0]}", f"BDG_HINT_GATHER={v[1]}", f"BDG_HINT_PREV={v[2]}", f"BDG_HINT_STORE={v[3]}"))
    return out
with ThreadPoolExecutor(4) as ex:
    for o in ex.map(one, variants):
        print(o)
