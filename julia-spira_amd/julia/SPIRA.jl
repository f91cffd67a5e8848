# SPIRA.jl — Julia host side of the MI355X path-trace backend (libspira_hip.so, C ABI in include/spira_hip.h).
#
# Keeps the surface of the reference package (src/SPIRA.jl:11-13, src/spira-metal-optimized.jl):
#   Camera(lookfrom, lookat, vup, vfov, aspect_ratio) / create_scene() / prepare_scene_data(scene) /
#   render(scene, camera, width, height; samples_per_pixel, max_depth, output_path) / render_hybrid_gpu
# and reaches the GPU through plain `ccall`s only — no Metal.jl, CUDA.jl, AMDGPU.jl or KernelAbstractions.
# NOT EXECUTED IN THIS PIPELINE: Julia is not installed in the build container or on the GPU box
# (SURVEY.md F2).  julia-spira_amd/spira_hip/spira.py is the executable twin that binds the SAME symbols
# with ctypes; tests/test_abi_cpu.py checks both against include/spira_hip.h by name.
module SPIRA

using StaticArrays

export Scene, Camera, Ray, Sphere, Material, Point3, Vec3, Color,
       render_hybrid_gpu, render_with_cpu, render, create_scene, prepare_scene_data

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

# mirrors of the C structs (include/spira_hip.h)
struct SpiraParams
    width::UInt32; height::UInt32; spp::UInt32; max_depth::UInt32
    n_spheres::UInt32; n_materials::UInt32; n_triangles::UInt32; flags::UInt32
    seed::UInt64
    row0::UInt32; rows::UInt32; stripe_h::UInt32; stripe_count::UInt32; stripe_rank::UInt32; batch_rays::UInt32
end

const SPIRA_POST_ACES_GAMMA = 0x00000100   # the display transform of gpu_tone_map_kernel! :1128-1144
const SPIRA_POST_NONE       = 0x00000300

spira_error(rc) = error("libspira_hip error $rc: " * unsafe_string(ccall((:spira_last_error, libspira), Cstring, ())))

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

# render_hybrid_gpu(width, height, scene, camera; samples_per_pixel, max_depth)  (:1228-1343)
# returns an H x W x 3 Float32 array, row 1 = image top (finalize_image_from_gpu_buffer :1157-1190);
# wrap with colorview(RGB, permutedims(img, (3, 1, 2))) where Images.jl is installed.
function render_hybrid_gpu(width::Int, height::Int, scene::Scene, camera::Camera;
                           samples_per_pixel::Int=16, max_depth::Int=4, seed::Integer=0, flags::Integer=SPIRA_POST_ACES_GAMMA)
    sphere_data, material_data = prepare_scene_data(scene)
    cam = Float32[camera.origin..., camera.lower_left_corner..., camera.horizontal..., camera.vertical...]
    p = Ref(SpiraParams(width, height, samples_per_pixel, max_depth, length(scene.spheres), length(scene.materials), 0,
                        UInt32(flags), UInt64(seed), 0, 0, 0, 0, 0, 0))
    planar = Array{Float32}(undef, width, height, 3)        # C order [3][H][W] == Julia (W, H, 3)
    rc = ccall((:spira_render_f32, libspira), Cint,
               (Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{SpiraParams}, Ptr{Float32}, Ptr{Float32}),
               sphere_data, material_data, C_NULL, cam, p, C_NULL, planar)
    rc == 0 || spira_error(rc)
    return permutedims(planar, (2, 1, 3))                   # (H, W, 3), row 1 = top
end

# render_with_cpu(width, height, scene, camera; ...) (:1346-1450, exported by src/SPIRA.jl:13): same estimator
# (trace_ray :1351-1412 = SPIRA_SEM_CPU) and display transform (clamp + sqrt :1441-1442), executed by the HIP kernels.
render_with_cpu(width::Int, height::Int, scene::Scene, camera::Camera; samples_per_pixel::Int=16, max_depth::Int=4, seed::Integer=0) =
    render_hybrid_gpu(width, height, scene, camera; samples_per_pixel=samples_per_pixel, max_depth=max_depth, seed=seed,
                      flags=0x00000001 | 0x00000200)        # SPIRA_SEM_CPU | SPIRA_POST_CLAMP_GAMMA

# render(scene, camera, width, height; samples_per_pixel=16, max_depth=4, output_path=...)  (:1453-1490)
function render(scene::Scene, camera::Camera, width::Int, height::Int;
                samples_per_pixel::Int=16, max_depth::Int=4, output_path::String="metal_optimized_render.png", seed::Integer=0)
    start_time = time()
    println("Rendering with HIP GPU (MI355X, GPU-side accumulation)...")
    img = render_hybrid_gpu(width, height, scene, camera; samples_per_pixel=samples_per_pixel, max_depth=max_depth, seed=seed)
    println("Render completed in $(round(time() - start_time, digits=2)) seconds")
    if !isempty(output_path) && Base.find_package("FileIO") !== nothing && Base.find_package("Images") !== nothing
        @eval using FileIO, Images        # image saving must not hard-depend on Images/FileIO (SURVEY.md §8b)
        Base.invokelatest(save, output_path, Base.invokelatest(colorview, Main.RGB, permutedims(img, (3, 1, 2))))
        println("Saved render to $output_path")
    end
    return img
end

end # module
