# SPIRA.jl — Julia host side of the MI355X path-trace backend (libspira_hip.so, C ABI in include/spira_hip.h).
#
# Keeps the surface of the reference package (src/SPIRA.jl:11-13, src/spira-metal-optimized.jl):
#   Camera(lookfrom, lookat, vup, vfov, aspect_ratio) / create_scene() / prepare_scene_data(scene) /
#   render(scene, camera, width, height; samples_per_pixel, max_depth, output_path) -> Matrix{RGB{Float32}} /
#   render_hybrid_gpu / render_with_cpu
# and reaches the GPU through plain `ccall`s only — no Metal.jl, CUDA.jl, AMDGPU.jl or KernelAbstractions.
# Dependencies: StaticArrays and Colors, both already dependencies of the reference (Project.toml:8,:21).
# Image files are written by this module itself (PNG, 32-bit EXR): Images / FileIO are NOT required.
#
# NOT EXECUTED IN THIS PIPELINE: Julia is not installed in the build container or on the GPU box
# (SURVEY.md F2).  julia-spira_amd/spira_hip/spira.py is the executable twin that binds the SAME symbols
# with ctypes; tests/test_abi_cpu.py parses every `ccall` below and checks its return and argument types
# against the prototypes of include/spira_hip.h, argument by argument.
module SPIRA

using StaticArrays
using Colors

export Scene, Camera, Ray, Sphere, Material, Point3, Vec3, Color,
       render_hybrid_gpu, render_with_cpu, render, render_multi, create_scene, prepare_scene_data,
       SceneHandle, destroy!, save_png, save_exr

const libspira = get(ENV, "SPIRA_HIP_LIB", joinpath(@__DIR__, "..", "csrc", "libspira_hip.so"))

const Vec3 = SVector{3, Float32}      # src/spira-metal-optimized.jl:282-284
const Point3 = Vec3
const Color = Vec3
const BLACK = Vec3(0f0, 0f0, 0f0)

struct Sphere                          # :305-311 (material: 1-based index)
    center::Point3
    radius::Float32
    material::Int
end

struct Material                        # :314-322
    albedo::Color
    emission::Color
    metallic::Float32
    roughness::Float32
    Material(albedo::Color; emission::Color=BLACK, metallic::Float32=0f0, roughness::Float32=0.5f0) =
        new(albedo, emission, metallic, roughness)
end

struct Ray                             # :293-298
    origin::Point3
    direction::Vec3
    Ray(o::Point3, d::Vec3) = new(o, d / sqrt(sum(d .* d)))
end

struct Scene                           # :351-354
    spheres::Vector{Sphere}
    materials::Vector{Material}
end

# mirror of spira_params (include/spira_hip.h, 64 bytes)
struct SpiraParams
    width::UInt32; height::UInt32; spp::UInt32; max_depth::UInt32
    n_spheres::UInt32; n_materials::UInt32; n_triangles::UInt32; flags::UInt32
    seed::UInt64
    row0::UInt32; rows::UInt32; stripe_h::UInt32; stripe_count::UInt32; stripe_rank::UInt32; batch_rays::UInt32
end

const SPIRA_SEM_A            = 0x00000000   # ray_color of examples/julia-raytracer.jl (the estimator parity is graded on)
const SPIRA_SEM_CPU          = 0x00000001   # trace_ray of render_with_cpu :1351-1412
const SPIRA_SEM_METAL        = 0x00000002   # path_trace of src/spira_path_trace_kernel.metal:140-269
const SPIRA_SEM_HYBRID       = 0x00000003   # the host loop of render_hybrid_gpu :1228-1343 itself, as written (last bounce shaded, tone map per sample)
const SPIRA_POST_ACES_GAMMA  = 0x00000100   # the display transform of gpu_tone_map_kernel! :1128-1144
const SPIRA_POST_CLAMP_GAMMA = 0x00000200   # clamp + sqrt of render_with_cpu :1441-1442
const SPIRA_POST_NONE        = 0x00000300

spira_error(rc) = error("libspira_hip error $rc: " * unsafe_string(ccall((:spira_last_error, libspira), Cstring, ())))

const SPIRA_ABI_VERSION = 3            # of the include/spira_hip.h these ccalls and SpiraParams were written against
function __init__()                    # a stale library (SPIRA_HIP_LIB, an old build) would read SpiraParams with another layout
    have = ccall((:spira_abi_version, libspira), Cint, ())
    have == SPIRA_ABI_VERSION || error("$libspira has ABI version $have, this module was written for $SPIRA_ABI_VERSION")
end

# which of the reference's estimators runs, and the display transform that goes with it (spira_hip/spira.py: SEMANTICS)
const SEMANTICS = Dict(:A => (SPIRA_SEM_A, SPIRA_POST_ACES_GAMMA), :cpu => (SPIRA_SEM_CPU, SPIRA_POST_CLAMP_GAMMA), :metal => (SPIRA_SEM_METAL, SPIRA_POST_ACES_GAMMA),
                       :hybrid => (SPIRA_SEM_HYBRID, SPIRA_POST_NONE))
semantics_flags(semantics::Symbol, flags) = flags === nothing ? (SEMANTICS[semantics][1] | SEMANTICS[semantics][2]) : Int(flags)

device_count() = Int(ccall((:spira_device_count, libspira), Cint, ()))
function set_device(d::Integer)
    rc = ccall((:spira_set_device, libspira), Cint, (Cint,), d)
    rc == 0 || spira_error(rc)
end

struct Camera                          # :325-348 — the arithmetic runs in spira_camera_lookat_f32
    origin::Point3
    lower_left_corner::Point3
    horizontal::Vec3
    vertical::Vec3
    function Camera(lookfrom::Point3, lookat::Point3, vup::Vec3, vfov::Real, aspect_ratio::Real)
        out = Vector{Float32}(undef, 12)
        rc = ccall((:spira_camera_lookat_f32, libspira), Cint,
                   (Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Cfloat, Ptr{Float32}),
                   collect(lookfrom), collect(lookat), collect(vup), Float32(vfov), Float32(aspect_ratio), out)
        rc == 0 || spira_error(rc)
        new(Vec3(out[1:3]), Vec3(out[4:6]), Vec3(out[7:9]), Vec3(out[10:12]))
    end
end

function create_scene()                # :429-510
    materials = [Material(Vec3(0.7f0, 0.3f0, 0.3f0); metallic=0f0, roughness=0.5f0),
                 Material(Vec3(0.5f0, 0.5f0, 0.5f0); metallic=0f0, roughness=0.9f0),
                 Material(Vec3(0.8f0, 0.8f0, 0.8f0); metallic=1f0, roughness=0f0),
                 Material(Vec3(0.8f0, 0.8f0, 1.0f0); metallic=0.9f0, roughness=0f0),
                 Material(Vec3(1f0, 1f0, 1f0); emission=Vec3(5f0, 5f0, 5f0), metallic=0f0, roughness=0f0)]
    spheres = [Sphere(Point3(0f0, 0f0, 0f0), 0.5f0, 1), Sphere(Point3(0f0, -100.5f0, 0f0), 100f0, 2),
               Sphere(Point3(1f0, 0f0, 0f0), 0.5f0, 3), Sphere(Point3(-1f0, 0f0, 0f0), 0.5f0, 4),
               Sphere(Point3(0f0, 5f0, 0f0), 1f0, 5)]
    return Scene(spheres, materials)
end

function prepare_scene_data(scene::Scene)   # :515-542 (flat Float32 arrays, material index stored as a float)
    sphere_data = zeros(Float32, 5 * length(scene.spheres))
    for (i, s) in enumerate(scene.spheres)
        k = (i - 1) * 5
        sphere_data[k+1:k+3] .= s.center; sphere_data[k+4] = s.radius; sphere_data[k+5] = Float32(s.material)
    end
    material_data = zeros(Float32, 8 * length(scene.materials))
    for (i, m) in enumerate(scene.materials)
        k = (i - 1) * 8
        material_data[k+1:k+3] .= m.albedo; material_data[k+4:k+6] .= m.emission
        material_data[k+7] = m.metallic; material_data[k+8] = m.roughness
    end
    return sphere_data, material_data
end

camera12(camera::Camera) = Float32[camera.origin..., camera.lower_left_corner..., camera.horizontal..., camera.vertical...]

make_params(width, height, spp, depth, scene::Scene, flags, seed) =
    SpiraParams(width, height, spp, depth, length(scene.spheres), length(scene.materials), 0, UInt32(flags), UInt64(seed), 0, 0, 0, 0, 0, 0)

# planar C output [3][H][W] (a Julia (W, H, 3) array), row 1 = image top  ->  Matrix{RGB{Float32}} (H x W), the type
# finalize_image_from_gpu_buffer returns (:1157-1190; the library has already applied its row flip :1177-1188)
function to_rgb_matrix(planar::Array{Float32,3})
    W, H, _ = size(planar)
    img = Matrix{RGB{Float32}}(undef, H, W)
    for j in 1:H, i in 1:W
        img[j, i] = RGB{Float32}(planar[i, j, 1], planar[i, j, 2], planar[i, j, 3])
    end
    return img
end

# ---- scene handles: upload (and, for meshes, build the BVH) once, render many frames  (replaces the per-render
# MtlArray(sphere_data) / MtlArray(material_data) of render_hybrid_gpu :1247-1254)
mutable struct SceneHandle
    ptr::Ptr{Cvoid}
    scene::Scene
    function SceneHandle(scene::Scene)
        sphere_data, material_data = prepare_scene_data(scene)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:spira_scene_create_f32, libspira), Cint,
                   (Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, UInt32, UInt32, UInt32, Ptr{Ptr{Cvoid}}),
                   sphere_data, material_data, C_NULL, length(scene.spheres), length(scene.materials), 0, out)
        rc == 0 || spira_error(rc)
        h = new(out[], scene)
        finalizer(destroy!, h)
        return h
    end
end

function destroy!(h::SceneHandle)
    if h.ptr != C_NULL
        ccall((:spira_scene_destroy, libspira), Cint, (Ptr{Cvoid},), h.ptr)
        h.ptr = C_NULL
    end
    return nothing
end

function render_hybrid_gpu(width::Int, height::Int, h::SceneHandle, camera::Camera;
                           samples_per_pixel::Int=16, max_depth::Int=4, seed::Integer=0, semantics::Symbol=:A, flags=nothing)
    p = Ref(make_params(width, height, samples_per_pixel, max_depth, h.scene, semantics_flags(semantics, flags), seed))
    planar = Array{Float32}(undef, width, height, 3)
    rc = ccall((:spira_render_scene_f32, libspira), Cint,
               (Ptr{Cvoid}, Ptr{Float32}, Ref{SpiraParams}, Ptr{Float32}, Ptr{Float32}),
               h.ptr, camera12(camera), p, C_NULL, planar)
    rc == 0 || spira_error(rc)
    return to_rgb_matrix(planar)
end

# render_hybrid_gpu(width, height, scene, camera; samples_per_pixel, max_depth)  (:1228-1343)
# -> Matrix{RGB{Float32}} (height x width, row 1 = image top), display transform of K7 (ACES + sqrt).
# `semantics` (:A default | :cpu | :metal) picks the estimator with its display transform; `flags` (SPIRA_SEM_* | SPIRA_POST_* | ...) overrides both.
function render_hybrid_gpu(width::Int, height::Int, scene::Scene, camera::Camera;
                           samples_per_pixel::Int=16, max_depth::Int=4, seed::Integer=0, semantics::Symbol=:A, flags=nothing)
    sphere_data, material_data = prepare_scene_data(scene)
    p = Ref(make_params(width, height, samples_per_pixel, max_depth, scene, semantics_flags(semantics, flags), seed))
    planar = Array{Float32}(undef, width, height, 3)        # C order [3][H][W] == Julia (W, H, 3)
    rc = ccall((:spira_render_f32, libspira), Cint,
               (Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{SpiraParams}, Ptr{Float32}, Ptr{Float32}),
               sphere_data, material_data, C_NULL, camera12(camera), p, C_NULL, planar)
    rc == 0 || spira_error(rc)
    return to_rgb_matrix(planar)
end

# The same frame on n_devices GPUs of this node: interleaved stripes, one RCCL gather to device 0 inside the library.
function render_multi(width::Int, height::Int, scene::Scene, camera::Camera, n_devices::Int=device_count();
                      samples_per_pixel::Int=16, max_depth::Int=4, seed::Integer=0, semantics::Symbol=:A, flags=nothing)
    sphere_data, material_data = prepare_scene_data(scene)
    p = Ref(make_params(width, height, samples_per_pixel, max_depth, scene, semantics_flags(semantics, flags), seed))
    planar = Array{Float32}(undef, width, height, 3)
    rc = ccall((:spira_render_multi_f32, libspira), Cint,
               (Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{SpiraParams}, Cint, Ptr{Float32}, Ptr{Float32}),
               sphere_data, material_data, C_NULL, camera12(camera), p, n_devices, C_NULL, planar)
    rc == 0 || spira_error(rc)
    return to_rgb_matrix(planar)
end

# render_with_cpu(width, height, scene, camera; ...) (:1346-1450, exported by src/SPIRA.jl:13): same estimator
# (trace_ray :1351-1412 = SPIRA_SEM_CPU) and display transform (clamp + sqrt :1441-1442), executed by the HIP kernels.
render_with_cpu(width::Int, height::Int, scene::Scene, camera::Camera; samples_per_pixel::Int=16, max_depth::Int=4, seed::Integer=0) =
    render_hybrid_gpu(width, height, scene, camera; samples_per_pixel=samples_per_pixel, max_depth=max_depth, seed=seed, semantics=:cpu)

# ---- image files without Images / FileIO (SURVEY.md 8b: "image saving must not hard-depend on Images/FileIO")
const CRC_TABLE = let t = Vector{UInt32}(undef, 256)
    for n in 0:255
        c = UInt32(n)
        for _ in 1:8
            c = (c & 1) != 0 ? (0xedb88320 ⊻ (c >> 1)) : (c >> 1)
        end
        t[n + 1] = c
    end
    t
end
function crc32(data::Vector{UInt8}, crc::UInt32=0x00000000)
    c = crc ⊻ 0xffffffff
    for b in data
        c = CRC_TABLE[((c ⊻ b) & 0xff) + 1] ⊻ (c >> 8)
    end
    return c ⊻ 0xffffffff
end
function adler32(data::Vector{UInt8})
    a, b = UInt32(1), UInt32(0)
    for x in data
        a = (a + x) % 65521
        b = (b + a) % 65521
    end
    return (b << 16) | a
end
be32(x) = UInt8[(x >> 24) & 0xff, (x >> 16) & 0xff, (x >> 8) & 0xff, x & 0xff]
function png_chunk(io::IO, tag::String, data::Vector{UInt8})
    body = vcat(Vector{UInt8}(tag), data)
    write(io, be32(UInt32(length(data))), body, be32(crc32(body)))
end

# 8-bit sRGB-less PNG of a display-referred image (values clamped to [0, 1]); deflate "stored" blocks, no zlib needed.
function save_png(path::String, img::Matrix{RGB{Float32}})
    H, W = size(img)
    raw = Vector{UInt8}(undef, H * (1 + 3W))
    k = 1
    q(v) = UInt8(round(clamp(v, 0f0, 1f0) * 255f0))
    for j in 1:H
        raw[k] = 0x00; k += 1                      # filter type 0
        for i in 1:W
            c = img[j, i]
            raw[k] = q(red(c)); raw[k+1] = q(green(c)); raw[k+2] = q(blue(c)); k += 3
        end
    end
    z = UInt8[0x78, 0x01]
    pos = 1
    while pos <= length(raw)
        n = min(65535, length(raw) - pos + 1)
        final = pos + n > length(raw) ? 0x01 : 0x00
        append!(z, UInt8[final, n & 0xff, (n >> 8) & 0xff, (~n) & 0xff, ((~n) >> 8) & 0xff])
        append!(z, @view raw[pos:pos+n-1])
        pos += n
    end
    append!(z, be32(adler32(raw)))
    open(path, "w") do io
        write(io, UInt8[0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a])
        png_chunk(io, "IHDR", vcat(be32(UInt32(W)), be32(UInt32(H)), UInt8[8, 2, 0, 0, 0]))
        png_chunk(io, "IDAT", z)
        png_chunk(io, "IEND", UInt8[])
    end
    return path
end

# save_exr(hdr_data, filename) of examples/julia-raytracer.jl:424-463 ("32-bit EXR"): scanline, uncompressed, FLOAT R/G/B.
function save_exr(hdr::Matrix{RGB{Float32}}, filename::String)
    H, W = size(hdr)
    attr(name, typ, data::Vector{UInt8}) = vcat(Vector{UInt8}(name), 0x00, Vector{UInt8}(typ), 0x00, reinterpret(UInt8, [Int32(length(data))]), data)
    le(x) = collect(reinterpret(UInt8, [x]))
    chlist = UInt8[]
    for n in ("B", "G", "R")
        append!(chlist, vcat(Vector{UInt8}(n), 0x00, le(Int32(2)), UInt8[0, 0, 0, 0], le(Int32(1)), le(Int32(1))))   # 2 = FLOAT
    end
    push!(chlist, 0x00)
    box = vcat(le(Int32(0)), le(Int32(0)), le(Int32(W - 1)), le(Int32(H - 1)))
    header = vcat(le(Int32(20000630)), le(Int32(2)),
                  attr("channels", "chlist", chlist), attr("compression", "compression", UInt8[0]),
                  attr("dataWindow", "box2i", box), attr("displayWindow", "box2i", box),
                  attr("lineOrder", "lineOrder", UInt8[0]), attr("pixelAspectRatio", "float", le(1f0)),
                  attr("screenWindowCenter", "v2f", vcat(le(0f0), le(0f0))), attr("screenWindowWidth", "float", le(1f0)), 0x00)
    line_bytes = 3 * W * 4
    data_pos = length(header) + 8 * H
    open(filename, "w") do io
        write(io, header)
        for y in 0:H-1
            write(io, UInt64(data_pos + y * (8 + line_bytes)))
        end
        for y in 1:H
            write(io, Int32(y - 1), Int32(line_bytes))
            write(io, Float32[blue(hdr[y, i]) for i in 1:W], Float32[green(hdr[y, i]) for i in 1:W], Float32[red(hdr[y, i]) for i in 1:W])
        end
    end
    println("Saved 32-bit EXR file: $filename")
    return true
end

# render(scene, camera, width, height; samples_per_pixel=16, max_depth=4, output_path=...)  (:1453-1490)
# `seed`, `semantics`, `flags` are additions (the reference never seeds and has one hard-wired estimator per entry point).
function render(scene::Scene, camera::Camera, width::Int, height::Int;
                samples_per_pixel::Int=16, max_depth::Int=4, output_path::String="metal_optimized_render.png", seed::Integer=0,
                semantics::Symbol=:A, flags=nothing)
    start_time = time()
    println("Rendering with HIP GPU (MI355X, GPU-side accumulation)...")
    img = render_hybrid_gpu(width, height, scene, camera; samples_per_pixel=samples_per_pixel, max_depth=max_depth, seed=seed,
                            semantics=semantics, flags=flags)
    println("Render completed in $(round(time() - start_time, digits=2)) seconds")
    if !isempty(output_path)                  # the reference calls FileIO.save (:1484); this module writes the file itself
        endswith(lowercase(output_path), ".exr") ? save_exr(img, output_path) : save_png(output_path, img)
        println("Saved render to $output_path")
    end
    return img
end

end # module
