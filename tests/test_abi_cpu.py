"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/spira_hip.h declares,
its host-side arithmetic agrees with the oracle, argument validation works, and — with no GPU —
render calls fail loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from spira_hip import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(binding):
    hdr = open(os.path.join(ROOT, "include", "spira_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(spira_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = binding.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "libspira_hip.so does not export %s" % name
    assert declared == set(binding.EXPORTS)
    assert lib.spira_abi_version() == 3


def test_struct_sizes(binding):
    assert C.sizeof(binding.Params) == 64
    assert C.sizeof(binding.Counters) == 15 * 8


def _header_struct_fields(name):
    """(type, field) pairs of `typedef struct <name> { ... } <name>;` in include/spira_hip.h, in declaration order."""
    hdr = open(os.path.join(ROOT, "include", "spira_hip.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        out += [(ctype, n.strip()) for n in names.split(",")]
    return out


def test_ctypes_structs_mirror_the_header_field_by_field(binding):
    """a struct that grows must bump SPIRA_ABI_VERSION, and the ctypes mirror must follow the header's fields in order,
    type and size — spira_get_counters writes sizeof(spira_counters) bytes into the caller's buffer."""
    ctype_of = {"uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "double": C.c_double}
    for struct, mirror in (("spira_params", binding.Params), ("spira_counters", binding.Counters)):
        fields = _header_struct_fields(struct)
        assert [n for _, n in fields] == [n for n, _ in mirror._fields_], struct
        assert [ctype_of[t] for t, _ in fields] == [t for _, t in mirror._fields_], struct
        assert C.sizeof(mirror) == sum(C.sizeof(ctype_of[t]) for t, _ in fields), struct      # no padding either side
    import oracle_py
    assert [n for _, n in _header_struct_fields("spira_params")] == [n for n, _ in oracle_py.SpiraParams._fields_]


def test_library_is_gfx950_only():
    import subprocess
    from spira_hip import _binding
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", "--input=" + _binding.LIB_PATH],
                         capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets
    else:   # fall back to scanning the fat binary for arch strings
        blob = open(_binding.LIB_PATH, "rb").read()
        assert b"gfx950" in blob and b"gfx942" not in blob and b"gfx90a" not in blob


def test_the_translation_units_hold_the_kernels_they_are_built_for():
    """csrc/Makefile builds spira_hip.hip three times: SPIRA_TU_F32 (-fno-slp-vectorize) must hold every Float32 kernel and no Float64 one,
    SPIRA_TU_F64MESH (compiler defaults) the Float64 path kernels of mesh scenes and nothing else, SPIRA_TU_MAIN (fewer branches folded into selects) every
    other Float64 kernel — a kernel instantiated in the wrong unit would silently run the slower code (profiles/r03_compiler_flags.md)."""
    import subprocess
    csrc = os.path.join(ROOT, "julia-spira_amd", "csrc")
    objs = {n: os.path.join(csrc, "spira_tu_%s.o" % n) for n in ("main", "f32", "f64mesh")}
    if not all(os.path.exists(o) for o in objs.values()):
        pytest.skip("objects not present (library built elsewhere)")
    syms = {n: subprocess.run(["nm", o], capture_output=True, text=True, check=True).stdout for n, o in objs.items()}
    count = lambda text, pat: len(re.findall(pat, text))
    # host stubs of the kernel templates: k_path<float ...> mangles to ...6k_pathIf..., k_path<double, R, BVH, ...> to ...6k_pathIdLi<R>ELb<BVH>E...
    for kern in ("8k_bounce", "6k_mega", "9k_resolve", "10k_finalize", "15k_variant_metal", "13k_variant_cpu", "12k_path_metal"):
        assert count(syms["main"], kern + "If") == 0 and count(syms["main"], kern + "Id") > 0, kern
        assert count(syms["f32"], kern + "Id") == 0 and count(syms["f32"], kern + "If") > 0, kern
        assert count(syms["f64mesh"], kern + "I[fd]") == 0, kern
    assert count(syms["f32"], "6k_pathIf") > 0 and count(syms["f32"], "6k_pathId") == 0
    assert count(syms["main"], "6k_pathIf") == 0 and count(syms["main"], "6k_pathIdLi[12]ELb0E") > 0 and count(syms["main"], "6k_pathIdLi[12]ELb1E") == 0
    assert count(syms["f64mesh"], "6k_pathIf") == 0 and count(syms["f64mesh"], "6k_pathIdLi[12]ELb1E") > 0 and count(syms["f64mesh"], "6k_pathIdLi[12]ELb0E") == 0
    for fn, unit in (("render_impl_f32", "f32"), ("trace_impl_f32", "f32"), ("launch_path_mesh_f64", "f64mesh"), ("launch_path_resume_f64", "f64mesh")):
        assert re.search(r" T .*%s" % fn, syms[unit]) and re.search(r" U .*%s" % fn, syms["main"]), fn
    assert not re.search(r" T spira_render_f32", syms["f32"]) and not re.search(r" T spira_render_f64", syms["f64mesh"])
    mk = open(os.path.join(csrc, "Makefile")).read()
    for flag in ("-DSPIRA_TU_F32", "-fno-slp-vectorize", "-DSPIRA_TU_MAIN", "-DSPIRA_TU_F64MESH", "-two-entry-phi-node-folding-threshold=1"):
        assert flag in mk, flag


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_camera_matches_oracle(binding, oracle, prec):
    rng = np.random.default_rng(1)
    for _ in range(20):
        pos, la = rng.normal(size=3) * 3, rng.normal(size=3)
        fov, asp = float(rng.uniform(20, 100)), float(rng.uniform(0.5, 2.5))
        a = binding.camera_lookat(pos, la, [0, 1, 0], fov, asp, 1.0, prec)
        b = oracle.camera(pos, la, [0, 1, 0], fov, asp, 1.0, prec)
        assert np.array_equal(a, b)
    c = binding.camera_lookat([0, 1, 3], [0, 0, 0], [0, 1, 0], 40.0, np.float32(16 / 9), prec="f32")
    assert np.allclose(c[3:6], [-0.6470582, 0.3384798, 2.1664143], atol=2e-6)   # SURVEY G7 KAT


def test_tonemap_matches_oracle(binding, oracle):
    x = np.concatenate([np.linspace(-1, 8, 500), [0.18, 1.0, 4.0]]).astype(np.float32)
    for post in (0x000, 0x100, 0x200, 0x300):
        got = binding.tonemap(x, post)
        want = np.array([oracle.post(float(v), post, "f32") for v in x], dtype=np.float32)
        assert np.array_equal(got, want)


def test_stripe_rows(binding):
    from spira_hip import distributed as D
    for h, sh, n in [(1080, 8, 8), (1080, 8, 3), (27, 4, 3), (17, 5, 4), (64, 64, 2)]:
        tot = 0
        for r in range(n):
            rows = binding.stripe_rows(h, sh, n, r)
            assert rows == len(D.rows_of_rank(h, n, r, sh))
            tot += rows
        assert tot == h
    assert binding.stripe_rows(100, 8, 0, 0) == 0 and binding.stripe_rows(100, 8, 2, 2) == 0


def test_validation_errors(binding):
    s = scenes.scene_s1()
    sp, ma, cam = s["spheres5"], s["materials8"], s["camera12"]

    def rc(params, spheres=sp, mats=ma):
        with pytest.raises(binding.SpiraError) as e:
            binding.render(spheres, mats, None, cam, params)
        return str(e.value)
    assert "error -1" in rc(binding.make_params(1, 10, 1, 1, 5, 5))                      # width < 2
    # the message is set inside render_impl<T> — for Float32 that is the library's OTHER translation unit — and read through spira_last_error():
    # both units must see one thread-local error string
    for prec in ("f32", "f64"):
        with pytest.raises(binding.SpiraError) as e:
            binding.render(sp, ma, None, cam, binding.make_params(1, 10, 1, 1, 5, 5), prec)
        assert "width and height must be >= 2" in str(e.value), prec
    assert "error -4" in rc(binding.make_params(8, 8, 0, 1, 5, 5))                       # spp = 0
    assert "error -4" in rc(binding.make_params(8, 8, 1, 256, 5, 5))                     # depth > 255
    assert "error -4" in rc(binding.make_params(8, 8, 1, 1, 2000, 5))                    # too many spheres
    bad = sp.copy()
    bad[2, 4] = 6
    assert "error -1" in rc(binding.make_params(8, 8, 1, 1, 5, 5), spheres=bad)          # material index out of range
    assert "error -5" in rc(binding.make_params(8, 8, 1, 1, 5, 5, flags=0x7))            # unknown semantics
    assert "error -1" in rc(binding.make_params(8, 8, 1, 1, 5, 5, row0=4, rows=8))       # tile outside the image
    assert "error -1" in rc(binding.make_params(8, 8, 1, 1, 5, 5, rows=3, stripe_h=2, stripe_count=2, stripe_rank=0))


def test_no_gpu_means_loud_failure(binding):
    """In the CPU container the product path must raise — never fall back to a CPU renderer."""
    if binding.device_count() > 0:
        pytest.skip("a HIP device is visible")
    s = scenes.scene_s1()
    with pytest.raises(binding.SpiraError) as e:
        binding.render(s["spheres5"], s["materials8"], None, s["camera12"], binding.make_params(16, 9, 1, 2, 5, 5))
    assert "error -2" in str(e.value)


def test_mirror_api_surface(binding):
    import spira_hip
    from spira_hip import raytracer
    for name in ["Scene", "Camera", "Ray", "Sphere", "Material", "render_hybrid_gpu", "render_with_cpu", "render", "create_scene",
                 "Point3", "Vec3", "Color"]:   # src/SPIRA.jl:11-13 + README.md:50-53
        assert hasattr(spira_hip, name)
    scene = spira_hip.create_scene()
    sd, md = spira_hip.prepare_scene_data(scene)
    s1 = scenes.scene_s1()
    assert sd.dtype == np.float32 and np.array_equal(sd.reshape(-1, 5), s1["spheres5"].astype(np.float32))
    assert np.array_equal(md.reshape(-1, 8), s1["materials8"].astype(np.float32))
    cam = spira_hip.Camera(spira_hip.Point3(0, 1, 3), spira_hip.Point3(0, 0, 0), spira_hip.Vec3(0, 1, 0), 40.0, 16 / 9)
    assert np.array_equal(cam.flat(), s1["camera12"].astype(np.float32))
    r = spira_hip.Ray(spira_hip.Point3(0, 0, 0), spira_hip.Vec3(0, 3, 4))
    assert np.allclose(r.direction, [0, 0.6, 0.8])
    world, camera = raytracer.create_scene()
    sp, ma, tr = raytracer.flatten_world(world)
    s2 = scenes.scene_s2()
    assert np.array_equal(sp, s2["spheres5"]) and np.array_equal(ma, s2["materials8"]) and np.array_equal(tr, s2["triangles10"])
    assert np.array_equal(camera.flat(), s2["camera12"])
    with pytest.raises(ValueError):
        raytracer.flatten_world(raytracer.BoundingVolumeHierarchy(list(reversed(world.objects))))


def test_png_writer(tmp_path):
    import zlib
    from spira_hip.png import save_png
    img = np.zeros((4, 5, 3), dtype=np.float32)
    img[1, 2] = [1.0, 0.5, 0.0]
    path = tmp_path / "x.png"
    save_png(str(path), img)
    blob = path.read_bytes()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n" and blob[12:16] == b"IHDR"
    i = blob.index(b"IDAT")
    n = int.from_bytes(blob[i - 4:i], "big")
    raw = zlib.decompress(blob[i + 4:i + 4 + n])
    assert len(raw) == 4 * (1 + 5 * 3) and raw[1 * 16 + 1 + 2 * 3:1 * 16 + 1 + 2 * 3 + 3] == bytes([255, 128, 0])


# --------------------------------------------------------------------- julia/SPIRA.jl `ccall`s against the header, argument by argument
_J2C = {   # Julia ccall type -> the C types it may stand for
    "Cint": {"int"}, "Cfloat": {"float"}, "Cdouble": {"double"}, "UInt32": {"uint32_t"}, "UInt64": {"uint64_t"},
    "Cstring": {"const char *"},
    "Ptr{Float32}": {"const float *", "float *"}, "Ptr{Float64}": {"const double *", "double *"},
    "Ptr{UInt32}": {"const uint32_t *", "uint32_t *"},
    "Ref{SpiraParams}": {"const spira_params *"},
    "Ptr{Cvoid}": {"void *", "spira_scene *", "const spira_scene *"}, "Ptr{Ptr{Cvoid}}": {"spira_scene **"},
}


def _c_prototypes():
    hdr = open(os.path.join(ROOT, "include", "spira_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[a-z_0-9]+\s*\**)\s*(spira_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr, flags=re.M):
        ctypes_ = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            m = re.match(r"^(.*?)([A-Za-z_][A-Za-z_0-9]*)?(\[\d*\])?$", a)
            typ, arr = m.group(1).strip(), m.group(3)
            if not typ:                       # unnamed parameter (e.g. "uint32_t")
                typ = m.group(2)
            if arr:
                typ += " *"
            typ = re.sub(r"\s*\*", " *", re.sub(r"\s+", " ", typ)).replace("* *", "**").strip()
            ctypes_.append(typ)
        protos[name] = (re.sub(r"\s*\*", " *", re.sub(r"\s+", " ", ret)).strip(), ctypes_)
    return protos


def _check_julia_ccalls(fname, min_calls):
    src = open(os.path.join(ROOT, "julia-spira_amd", "julia", fname)).read()
    src = re.sub(r"#.*", "", src)
    protos = _c_prototypes()
    calls = re.findall(r"ccall\(\(:(spira_[a-z0-9_]+),\s*libspira\),\s*([A-Za-z0-9{}]+),\s*\(([^()]*)\)", src)
    assert len(calls) >= min_calls, fname
    seen = set()
    for name, ret, args in calls:
        assert name in protos, "%s calls %s, which include/spira_hip.h does not declare" % (fname, name)
        cret, cargs = protos[name]
        assert cret in _J2C[ret], (name, ret, cret)
        jargs = [a.strip() for a in args.split(",") if a.strip()]
        assert len(jargs) == len(cargs), (name, jargs, cargs)
        for k, (ja, ca) in enumerate(zip(jargs, cargs)):
            assert ca in _J2C[ja], "%s: %s argument %d: Julia %s vs C %s" % (fname, name, k + 1, ja, ca)
        seen.add(name)
    # the struct mirror: field order and widths of SpiraParams == spira_params
    fields = re.search(r"struct SpiraParams(.*?)\nend", src, flags=re.S).group(1)
    jf = re.findall(r"([a-z_0-9]+)::(UInt32|UInt64)", fields)
    from spira_hip import _binding
    assert [(n, {"UInt32": 4, "UInt64": 8}[t]) for n, t in jf] == [(n, C.sizeof(t)) for n, t in _binding.Params._fields_], fname
    # flag constants used by the shim
    hdr = open(os.path.join(ROOT, "include", "spira_hip.h")).read()
    for const, val in re.findall(r"const (SPIRA_[A-Z_]+)\s*=\s*(0x[0-9a-fA-F]+)", src):
        m = re.search(r"#define %s\s+(0x[0-9a-fA-F]+)u" % const, hdr)
        assert m and int(m.group(1), 16) == int(val, 16), (fname, const)
    return seen


def test_julia_oracle_script_surface_ccalls_match_header():
    """julia/Raytracer.jl = the surface of examples/julia-raytracer.jl (Float64, triangles) over the same ABI."""
    seen = _check_julia_ccalls("Raytracer.jl", 3)
    assert {"spira_render_f64", "spira_camera_lookat_f64", "spira_last_error"} <= seen


def test_julia_ccalls_match_header_argument_by_argument():
    src = open(os.path.join(ROOT, "julia-spira_amd", "julia", "SPIRA.jl")).read()
    src = re.sub(r"#.*", "", src)
    protos = _c_prototypes()
    calls = re.findall(r"ccall\(\(:(spira_[a-z0-9_]+),\s*libspira\),\s*([A-Za-z0-9{}]+),\s*\(([^()]*)\)", src)
    assert len(calls) >= 9
    seen = set()
    for name, ret, args in calls:
        assert name in protos, "SPIRA.jl calls %s, which include/spira_hip.h does not declare" % name
        cret, cargs = protos[name]
        assert cret in _J2C[ret], (name, ret, cret)
        jargs = [a.strip() for a in args.split(",") if a.strip()]
        assert len(jargs) == len(cargs), (name, jargs, cargs)
        for k, (ja, ca) in enumerate(zip(jargs, cargs)):
            assert ca in _J2C[ja], "%s argument %d: Julia %s vs C %s" % (name, k + 1, ja, ca)
        seen.add(name)
    assert {"spira_render_f32", "spira_render_scene_f32", "spira_render_multi_f32", "spira_scene_create_f32", "spira_scene_destroy",
            "spira_camera_lookat_f32", "spira_last_error", "spira_device_count", "spira_set_device"} <= seen
    # the struct mirror: field order and widths of SpiraParams == spira_params
    fields = re.search(r"struct SpiraParams(.*?)\nend", src, flags=re.S).group(1)
    jf = re.findall(r"([a-z_0-9]+)::(UInt32|UInt64)", fields)
    from spira_hip import _binding
    assert [(n, {"UInt32": 4, "UInt64": 8}[t]) for n, t in jf] == [(n, C.sizeof(t)) for n, t in _binding.Params._fields_]
    # flag constants used by the shim
    hdr = open(os.path.join(ROOT, "include", "spira_hip.h")).read()
    for const, val in re.findall(r"const (SPIRA_[A-Z_]+)\s*=\s*(0x[0-9a-fA-F]+)", src):
        m = re.search(r"#define %s\s+(0x[0-9a-fA-F]+)u" % const, hdr)
        assert m and int(m.group(1), 16) == int(val, 16), const


def _julia_exports(fname):
    src = open(os.path.join(ROOT, "julia-spira_amd", "julia", fname)).read()
    m = re.search(r"^export (.*?)\n\n", src, flags=re.S | re.M)
    return set(x.strip() for x in m.group(1).replace("\n", " ").split(",") if x.strip()), src


def test_julia_export_lists_cover_the_reference_and_the_python_twins():
    """Each Julia module exports what the reference surface it stands in for exports / defines, and what its executable Python twin
    offers (a maintainer swapping the module into tests/bunny-test.jl must not meet an UndefVarError)."""
    ex, src = _julia_exports("SPIRA.jl")
    # /root/reference/src/SPIRA.jl:11-13
    assert {"Scene", "Camera", "Material", "Sphere", "Ray", "render", "create_scene", "render_hybrid_gpu", "render_with_cpu"} <= ex
    from spira_hip import spira
    twin = {n for n in ("Scene", "Camera", "Material", "Sphere", "Ray", "render", "create_scene", "prepare_scene_data", "render_hybrid_gpu", "render_with_cpu")
            if hasattr(spira, n)}
    assert twin <= ex, twin - ex
    for kw in ("semantics::Symbol=:A", "flags=nothing"):      # the estimator can be chosen from Julia too (spira.py: semantics=)
        assert kw in src[src.index("function render(scene::Scene"):], kw
    assert set(re.findall(r":(A|cpu|metal|hybrid) =>", src)) == set(spira.SEMANTICS) == {"A", "cpu", "metal", "hybrid"}
    ex, src = _julia_exports("Raytracer.jl")
    # the functions of /root/reference/examples/julia-raytracer.jl that tests/bunny-test.jl:37-60 and the script's own main() call
    assert {"Vec3", "Ray", "Material", "Sphere", "Triangle", "Mesh", "HittableList", "BoundingVolumeHierarchy", "Camera", "render", "to_acescg", "save_exr",
            "load_obj_mesh", "create_scene", "create_scene_with_obj", "render_example"} <= ex
    from spira_hip import raytracer
    twin = {n for n in ("Vec3", "Material", "Sphere", "Triangle", "BoundingVolumeHierarchy", "Camera", "render", "to_acescg", "save_exr", "load_obj_mesh",
                        "create_scene", "create_scene_with_obj", "flatten_world") if hasattr(raytracer, n)}
    assert twin <= ex, twin - ex
    for fn in ("function load_obj_mesh(filename::String, material::Material;", "function create_scene()", "function create_scene_with_obj(", "function render_example(;"):
        assert fn in src, fn
    # the vertex pipeline in the reference's order (:510-591): centre, normalise, rotate X / Y / Z, scale, translate
    body = src[src.index("function load_obj_mesh"):src.index("example_camera() =")]
    order = [body.index(k) for k in ("center_point", "1.0 / max_dimension", "rotation.x != 0\n", "rotation.y != 0\n", "rotation.z != 0\n", "v.x * scale.x", "v + translation")]
    assert order == sorted(order)


def test_both_julia_modules_refuse_a_library_of_another_abi_version():
    for fname in ("SPIRA.jl", "Raytracer.jl"):
        src = open(os.path.join(ROOT, "julia-spira_amd", "julia", fname)).read()
        assert "const SPIRA_ABI_VERSION = 3" in src and "function __init__()" in src and "ccall((:spira_abi_version, libspira), Cint, ())" in src, fname
    from spira_hip import _binding
    assert _binding.ABI_VERSION == 3


def test_flatten_world_gives_one_material_row_per_distinct_material(binding):
    """A mesh shares one material (examples/julia-raytracer.jl:598): both hosts must give it ONE row of the table — a row per triangle
    would need n x 64 bytes of LDS and fail with SPIRA_E_LIMIT above ~1 900 triangles."""
    from spira_hip import raytracer as R
    m = R.Material(diffuse=R.Vec3(0.7, 0.3, 0.2), specular=0.2, roughness=0.4)
    tris = [R.Triangle([R.Vec3(i, 0, 0), R.Vec3(i + 1, 0, 0), R.Vec3(i, 1, 0)], m) for i in range(5000)]
    world = R.BoundingVolumeHierarchy([R.Sphere(R.Vec3(0, -100.5, -1), 100, R.Material(diffuse=R.Vec3(0.8, 0.8, 0.2)))] + tris)
    spheres5, materials8, triangles10 = R.flatten_world(world)
    assert len(materials8) == 2 and len(triangles10) == 5000 and set(triangles10[:, 9]) == {2.0}
    src = open(os.path.join(ROOT, "julia-spira_amd", "julia", "Raytracer.jl")).read()
    body = src[src.index("function flatten_world"):src.index("aces1(x)")]
    assert "Dict{Material,Int}()" in body and "get!(mat_index, m) do" in body      # the Julia twin dedupes by value (it cannot run here)


def test_julia_png_and_exr_writers_mirror_the_python_ones(tmp_path):
    """The Julia file writers cannot run here; their byte layout is the Python writers' (spira_hip/png.py, exr.py), which CAN:
    PNG with stored deflate blocks decodes back, the EXR header matches the fields the Julia code emits."""
    import struct
    import zlib
    from spira_hip import exr, png
    rng = np.random.default_rng(0)
    img = rng.random((5, 7, 3)).astype(np.float32)
    p = str(tmp_path / "a.png")
    png.save_png(p, img)
    blob = open(p, "rb").read()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n" and b"IHDR" in blob and b"IEND" in blob
    i = blob.index(b"IDAT")
    n = struct.unpack(">I", blob[i - 4:i])[0]
    raw = zlib.decompress(blob[i + 4:i + 4 + n])
    assert len(raw) == 5 * (1 + 3 * 7)
    e = str(tmp_path / "a.exr")
    exr.save_exr(e, img)
    head = open(e, "rb").read(400)
    for key in (b"channels\0chlist", b"compression\0compression", b"dataWindow\0box2i", b"lineOrder\0lineOrder", b"screenWindowWidth\0float"):
        assert key in head
    src = open(os.path.join(ROOT, "julia-spira_amd", "julia", "SPIRA.jl")).read()
    for key in ("channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio", "screenWindowCenter", "screenWindowWidth", "20000630"):
        assert key in src


def test_multi_device_stripe_logic_on_cpu(binding):
    """The tiling arithmetic of spira_render_multi_* (host side, no GPU): spira_stripe_rows agrees with the Python sharding for every
    (height, devices, stripe height), and the row mapping k_assemble uses on device 0 — global row y lives at rank (y / h) % n, local
    row ((y / h) / n) * h + y % h — is the inverse of the ranks' local row order."""
    from spira_hip import distributed as D
    for H in (1, 7, 8, 9, 27, 117, 200, 1080):
        for n in (1, 2, 3, 5, 8):
            for h in (1, 2, 4, 8):
                rows = [D.rows_of_rank(H, n, r, h) for r in range(n)] if n > 1 else [list(range(H))]
                for r in range(n):
                    assert binding.stripe_rows(H, h, n, r) == len(rows[r]), (H, n, h, r)
                sr, sl = D.source_of_rows(H, n, h) if n > 1 else (np.zeros(H, int), np.arange(H))
                for y in range(H):
                    sq = y // h
                    rank, local = (sq % n, (sq // n) * h + y % h)
                    assert (rank, local) == (int(sr[y]), int(sl[y])) and rows[rank][local] == y
    assert binding.stripe_rows(100, 8, 0, 0) == 0 and binding.stripe_rows(100, 8, 4, 4) == 0     # bad rank / count
