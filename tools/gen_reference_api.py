#!/usr/bin/env python3
"""Lists what a crate OUTSIDE the reference can name and what its Participant offers: the public module paths and top-level public
items of AlexiaChen/mpvss-rs (src/lib.rs and the files of its `pub mod`s) and, for every hand-specialised
`impl Participant<...Group>` block of src/participant.rs, the public methods with their parameter names -- NAMES ONLY, no source
text.  The reference does not travel to the GPU box, so the listing is committed (tests/reference_api/reference_api.json) and
tests/test_capi_host.py holds rust/ against it: every `mpvss_rs::` path rust/ uses must be public (round 5 used the private
`mpvss_rs::util`), and the Participant of rust/src/participant.rs must offer the same methods with the same arity.
Run in the build container:  python3 tools/gen_reference_api.py [/root/reference] > tests/reference_api/reference_api.json"""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = os.path.join(ref, "src")


def read(rel):
    with open(os.path.join(src, rel)) as fh:
        return fh.read()


def top_level_pub_items(text):
    return sorted(set(m.group(2) for m in re.finditer(r"^pub (struct|trait|type|fn|enum|const|static) ([A-Za-z_0-9]+)", text, re.M)))


lib = read("lib.rs")
pub_mods = sorted(re.findall(r"^pub mod ([a-z_0-9]+);", lib, re.M))
private_mods = sorted(re.findall(r"^mod ([a-z_0-9]+);", lib, re.M))
paths = set("mpvss_rs::" + m for m in pub_mods)
for name in top_level_pub_items(lib):
    paths.add("mpvss_rs::" + name)
for m in re.finditer(r"^pub use ([a-z_0-9:]+)::\{?([A-Za-z_0-9, ]+)\}?;", lib, re.M):
    for name in m.group(2).split(","):
        paths.add("mpvss_rs::" + name.strip())
for mod in pub_mods:
    rel = mod + ".rs" if os.path.exists(os.path.join(src, mod + ".rs")) else os.path.join(mod, "mod.rs")
    text = read(rel)
    for name in top_level_pub_items(text):
        paths.add(f"mpvss_rs::{mod}::{name}")
    for m in re.finditer(r"^pub use ([a-z_0-9:]+)::\{?([A-Za-z_0-9, ]+)\}?;", text, re.M):
        for name in m.group(2).split(","):
            paths.add(f"mpvss_rs::{mod}::{name.strip()}")

part = read("participant.rs")
blocks = {}
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^impl(?:<[^>]*>)? Participant<([A-Za-z0-9_]+)>", part, re.M)]
for k, (pos, who) in enumerate(starts):
    end = starts[k + 1][0] if k + 1 < len(starts) else len(part)
    body = part[pos:end]
    cut = body.find("#[cfg(test)]")
    if cut >= 0:
        body = body[:cut]
    methods = {}
    for m in re.finditer(r"^    pub fn ([a-z_0-9]+)\s*\(([^)]*)\)", body, re.M | re.S):
        params = [p.strip().split(":")[0].strip() for p in m.group(2).split(",") if p.strip()]
        methods[m.group(1)] = {"receiver": params[0] if params and "self" in params[0] else None,
                               "params": [p for p in params if "self" not in p]}
    blocks.setdefault(who, {}).update(methods)
fields = re.search(r"pub struct Participant<G: Group> \{(.*?)\}", part, re.S).group(1)
json.dump({"crate": "mpvss-rs", "version": re.search(r'^version = "([^"]+)"', open(os.path.join(ref, "Cargo.toml")).read(), re.M).group(1),
           "public_paths": sorted(paths), "private_modules": private_mods,
           "participant_fields": {"public": re.findall(r"pub ([a-z_]+):", fields), "private": re.findall(r"^\s+([a-z_]+):", fields, re.M)},
           "participant_impls": blocks}, sys.stdout, indent=1, sort_keys=True)
print()
