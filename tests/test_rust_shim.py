"""The Rust binding a maintainer of the reference adds (integration/ffi.rs, integration/render_gpu.rs) cannot
be compiled here (no rustc), so it is checked as text against include/rrt.h: every function of the header
is declared with the same name and argument count, every #[repr(C)] struct has the header's fields in the
header's order with types of the same size, sizes and offsets agree with what gcc computes for the C
structs, the constants have the header's values, and render_gpu.rs only calls functions and names
struct fields that ffi.rs declares."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUST_SIZES = {"u8": 1, "i8": 1, "u32": 4, "i32": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "c_int": 4}
C_SIZES = {"uint8_t": 1, "uint32_t": 4, "int32_t": 4, "float": 4, "uint64_t": 8, "int64_t": 8, "double": 8, "int": 4}
STRUCTS = {"RrtMaterial": "rrt_material", "RrtLight": "rrt_light", "RrtRect": "rrt_rect", "RrtPlane": "rrt_plane",
           "RrtSphere": "rrt_sphere", "RrtSceneDesc": "rrt_scene_desc", "RrtSettings": "rrt_settings", "RrtStats": "rrt_stats"}


def _header():
    text = open(os.path.join(ROOT, "include", "rrt.h")).read()
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _rust():
    text = open(os.path.join(ROOT, "integration", "ffi.rs")).read()
    return re.sub(r"//[^\n]*", "", text)


def _c_structs(text):
    """name -> [(field, size, count)] for every `typedef struct name { ... } name;`."""
    out = {}
    for name, body in re.findall(r"typedef struct (\w+) \{(.*?)\} \1;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl, flags=re.S)
            ctype, ptr, rest = m.group(2), m.group(3), m.group(4)
            for item in rest.split(","):
                item = item.strip()
                star = ptr or ("*" if item.startswith("*") else "")
                item = item.lstrip("* ")
                am = re.match(r"(\w+)(?:\[(\d+)\])?$", item)
                fields.append((am.group(1), 8 if star else C_SIZES[ctype], int(am.group(2) or 1)))
        out[name] = fields
    return out


def _rust_structs(text):
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\][^{]*?pub struct (\w+) \{(.*?)\}", text, flags=re.S):
        fields = []
        for fname, ftype in re.findall(r"pub (\w+): ([^,\n]+),", body):
            ftype = ftype.strip()
            am = re.match(r"\[(\w+); (\d+)\]$", ftype)
            if am:
                fields.append((fname, RUST_SIZES[am.group(1)], int(am.group(2))))
            elif ftype.startswith("*const") or ftype.startswith("*mut"):
                fields.append((fname, 8, 1))
            else:
                fields.append((fname, RUST_SIZES[ftype], 1))
        out[name] = fields
    return out


def _layout(fields):
    """repr(C) layout: offsets and total size (arrays align as their element)."""
    off, offsets, align_max = 0, [], 1
    for _, size, count in fields:
        off = (off + size - 1) // size * size
        offsets.append(off)
        off += size * count
        align_max = max(align_max, size)
    return offsets, (off + align_max - 1) // align_max * align_max


def test_every_function_of_the_header_is_declared_in_ffi_rs():
    header, rust = _header(), _rust()
    c_fns = {}
    for name, args in re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header):
        args = args.strip()
        c_fns[name] = 0 if args in ("", "void") else len(args.split(","))
    block = re.search(r'extern "C" \{(.*?)\n\}', rust, flags=re.S).group(1)
    r_fns = {}
    for name, args in re.findall(r"pub fn (rrt_\w+)\((.*?)\)(?:\s*->\s*[^;]+)?;", block, flags=re.S):
        r_fns[name] = len([a for a in args.split(",") if a.strip()])
    assert set(c_fns) == set(r_fns), set(c_fns) ^ set(r_fns)
    assert c_fns == r_fns, {k: (c_fns[k], r_fns[k]) for k in c_fns if c_fns[k] != r_fns[k]}
    assert len(c_fns) >= 26


def test_repr_c_structs_match_the_header(tmp_path):
    c_structs, r_structs = _c_structs(_header()), _rust_structs(_rust())
    lines = []
    for rname, cname in STRUCTS.items():
        assert rname in r_structs, rname
        cf, rf = c_structs[cname], r_structs[rname]
        assert [f[0] for f in cf] == [f[0] for f in rf], (cname, [f[0] for f in cf], [f[0] for f in rf])
        assert [(f[1], f[2]) for f in cf] == [(f[1], f[2]) for f in rf], cname
        lines.append('printf("%%zu", sizeof(%s));' % cname)
        lines += ['printf(" %%zu", offsetof(%s, %s));' % (cname, f[0]) for f in cf]
        lines.append('printf("\\n");')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rrt.h"\nint main(void){%s return 0;}\n' % "".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    rows = [[int(v) for v in line.split()] for line in subprocess.check_output([str(exe)]).decode().splitlines()]
    for (rname, cname), row in zip(STRUCTS.items(), rows):
        offsets, size = _layout(r_structs[rname])
        assert row[0] == size, (cname, row[0], size)
        assert row[1:] == offsets, (cname, row[1:], offsets)


def test_constants_have_the_header_values():
    header, rust = _header(), _rust()
    consts = {}
    for body in re.findall(r"enum \{(.*?)\};", header, flags=re.S):
        for name, value in re.findall(r"(RRT_\w+) = (-?\d+)", body):
            consts[name] = int(value)
    for name, value in re.findall(r"#define (RRT_\w+) \(?(-?\d+)u?\)?", header):
        consts[name] = int(value)
    consts.pop("RRT_H", None)
    r_consts = {name: int(value) for name, value in re.findall(r"pub const (RRT_\w+): \w+ = (-?\d+);", rust)}
    assert consts == r_consts, {k: (consts.get(k), r_consts.get(k)) for k in set(consts) | set(r_consts) if consts.get(k) != r_consts.get(k)}
    assert len(consts) >= 25


def test_render_gpu_rs_uses_only_what_ffi_rs_declares():
    rust = _rust()
    shim = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "integration", "render_gpu.rs")).read())
    declared = set(re.findall(r"pub fn (\w+)", rust)) | set(re.findall(r"pub const (\w+)", rust)) | set(_rust_structs(rust)) | {"RrtCtx"}
    used = set(re.findall(r"ffi::(\w+)", shim))
    assert used <= declared, used - declared
    assert {"rrt_create", "rrt_upload_scene", "rrt_set_camera", "rrt_film_alloc", "rrt_set_filter", "rrt_render",
            "rrt_film_download", "rrt_destroy"} <= used
    r_structs = _rust_structs(rust)
    # every struct literal names exactly the struct's fields
    for name, body in re.findall(r"(?<!-> )ffi::(Rrt\w+) \{(.*?)\}", shim, flags=re.S):
        given = re.findall(r"(\w+):", re.sub(r"\([^()]*\)", "", body))  # (field: value pairs; nested calls removed)
        shorthand = [w for w in re.findall(r"(?:^|,)\s*(\w+)\s*(?=,)", body)]
        names = set(given) | set(shorthand)
        expected = {f[0] for f in r_structs[name]}
        assert expected <= names, (name, expected - names)
    # no name is used before it exists: the identifiers the earlier draft left undefined
    for ghost in ("r2c", "c2w", " x0", " y1"):
        assert ghost not in shim
