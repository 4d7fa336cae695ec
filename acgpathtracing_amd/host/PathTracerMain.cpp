// PathTracerMain.cpp — headless counterpart of the reference's PathTracer_Optix/PathTracerMain.cpp.
//
// Same structure, same function names over a PathTracerState, same defaults (512x512, 128 samples
// per launch, maxDepth 4, both toggles off, the hard-coded area light and camera); what the
// reference hard-codes or reads from the keyboard is a command line here, and frames go to image
// files instead of a GLFW window:
//
//   acgpt_main --obj scene.obj [--width 512 --height 512] [--spp-per-launch 128] [--frames 8] [--fuse-frames 1]
//              [--max-depth 4] [--direct-lighting] [--importance-sampling] [--device 0]
//              [--keys "0,1,UP,UP,R"] [--out frame.png] [--dump-every k] [--zero-copy]
//              [--orbit dx,dy] [--zoom n] [--sample-chunks c] [--build-mode 0|1]
//              [--gpus N] [--multi] [--save-accum file] [--restore-accum file] [--light-mode 0|1] [--math fast|ieee]
//
// --math: arithmetic of the shading code (pt_set_math_mode).  fast (default) is what the reference's own build computes with —
// nvcc --use_fast_math, /root/reference/CMakeLists.txt:267 —, ieee the correctly rounded level of the CPU oracle.
//
// --gpus N renders on devices 0..N-1 of the node through ONE context (pt_create_multi): pixel tiles of
// sutil/WorkDistribution.h per device, one RCCL reduce of the accumulation per launch — the reference's dormant multi-GPU
// path, behind the same functions.  --save-accum / --restore-accum write and read the progressive state of the reference
// (params.accumulationBuffer + currentFrameIdx, pathTracerPrograms.cu:803-811): a restored run continues the running mean
// where the saved one stopped, bit for bit.
//
// --keys replays the reference's key handler (PathTracerMain.cpp:100-141) between frames, one key
// per frame: 0 = direct lighting, 1 = importance sampling, UP/DOWN = max depth +-1 in [1,28],
// R = reset, Q = quit; every change resets the accumulation and the frame index.
// --orbit / --zoom drive the camera through Trackball (sutil/Trackball.cpp:51-137: 0.5 degree per
// pixel, zoom factor 1.1 per step) before the first frame; the reference includes Trackball but never
// wires it to an input callback (PathTracerMain.cpp:18, 686-688).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/acgpt.h"
#include "Camera.h"
#include "Exception.h"
#include "ImageIO.h"
#include "OutputBuffer.h"
#include "TinyObjWrapper.h"
#include "Trackball.h"

using namespace acgpt;

constexpr unsigned int maxiumumRecursionDepth = 28;     // PathTracerMain.cpp:42
static int32_t samples_per_launch = 128;                // :43
static uint32_t frame_counter = 0, sample_summ = 0;
static double avg_ms = 0.0, total_ms = 0.0;
static bool refreshAccumulationBuffer = false;
static Camera g_camera;

struct PathTracerState {                                // :71-93
    pt_ctx* context = nullptr;
    pt_params params = {};
    int device = 0;
    int gpus = 1;                                       // > 1: one context over devices 0..gpus-1
    bool multi = false;                                 // --multi: the group context even for one device (its RCCL reduce then runs with one rank)
};

static bool keyCallback(PathTracerState& state, const std::string& key)   // :100-141; false = quit
{
    pt_params* params = &state.params;
    if (key == "Q" || key == "ESC") return false;
    if (key == "0") {
        params->useDirectLighting = !params->useDirectLighting;
        std::cout << std::endl << "Using Direct Lighting: " << (params->useDirectLighting ? "yes" : "no") << std::endl;
        refreshAccumulationBuffer = true;
    } else if (key == "1") {
        params->useImportanceSampling = !params->useImportanceSampling;
        std::cout << std::endl << "Using Importance Sampling: " << (params->useImportanceSampling ? "yes" : "no") << std::endl;
        refreshAccumulationBuffer = true;
    } else if (key == "UP") {
        params->maxDepth = std::min((int)maxiumumRecursionDepth, (int)params->maxDepth + 1);
        refreshAccumulationBuffer = true;
        std::cout << std::endl << "Max Depth: " << params->maxDepth << std::endl;
    } else if (key == "DOWN") {
        params->maxDepth = std::max(1, (int)params->maxDepth - 1);
        refreshAccumulationBuffer = true;
        std::cout << std::endl << "Max Depth: " << params->maxDepth << std::endl;
    } else if (key == "R") {
        refreshAccumulationBuffer = true;
    }
    return true;
}

static void allocAccumulation(PathTracerState& state)
{
    void* p = nullptr;
    PT_CHECK(state.context, pt_device_malloc(state.context, &p, (size_t)state.params.width * state.params.height * 4 * sizeof(float)));
    state.params.accumulationBuffer = (float*)p;
    // zero-filled: with pt_set_partition(rank, world) the pixels of other ranks are never written and the
    // cross-rank reduce(SUM) relies on them being 0 (the reference's single-GPU cudaMalloc leaves them undefined)
    PT_CHECK(state.context, pt_device_memset(state.context, p, 0, (size_t)state.params.width * state.params.height * 4 * sizeof(float)));
}

static void initializeTheLaunch(PathTracerState& state)                   // :143-164
{
    allocAccumulation(state);
    state.params.frameBuffer = nullptr;
    state.params.samplesPerPixel = samples_per_launch;
    state.params.currentFrameIdx = 0u;
    pt_area_light& al = state.params.areaLight;
    al.emission = {10.0f, 10.0f, 10.0f};
    al.corner = {343.0f, 547.0f, 227.0f};
    al.v1 = {0.0f, 0.0f, 105.0f};
    al.v2 = {-130.0f, 0.0f, 0.0f};
    const float3 n = normalize(cross(make_float3(al.v1.x, al.v1.y, al.v1.z), make_float3(al.v2.x, al.v2.y, al.v2.z)));
    al.normal = {n.x, n.y, n.z};
    state.params.handle = pt_scene_handle(state.context);
}

static void updateState(OutputBuffer<uchar4>&, PathTracerState& state)    // :166-182
{
    if (refreshAccumulationBuffer) {
        refreshAccumulationBuffer = false;
        state.params.currentFrameIdx = 0;
        sample_summ = 0; frame_counter = 0; avg_ms = 0; total_ms = 0;
        PT_CHECK(state.context, pt_device_free(state.context, state.params.accumulationBuffer));
        allocAccumulation(state);
    }
}

// sub_frames > 1 (--fuse-frames): that many consecutive launches of the reference's loop in one kernel launch
static void LaunchCurrentFrame(OutputBuffer<uchar4>& output_buffer, PathTracerState& state, uint32_t sub_frames = 1)   // :184-210
{
    uchar4* result_buffer_data = output_buffer.map();
    state.params.frameBuffer = reinterpret_cast<uint8_t*>(result_buffer_data);
    PT_CHECK(state.context, pt_launch_frames(state.context, &state.params, sub_frames));
    output_buffer.unmap();
}

static void initCamera()                                                 // :228-233
{
    g_camera.setEye(make_float3(278.0f, 273.0f, -900.0f));
    g_camera.setLookat(make_float3(278.0f, 273.0f, 330.0f));
    g_camera.setUp(make_float3(0.0f, 1.0f, 0.0f));
    g_camera.setFovY(35.0f);
}

static void createDeviceContext(PathTracerState& state)                  // :240-258
{
    if (state.gpus > 1 || state.multi) {
        // devices device, device + 1, ...; as a rehearsal on a one-GPU box (ACGPT_REHEARSE_SAME_GPU=1) every rank shares `device`
        const char* reh = getenv("ACGPT_REHEARSE_SAME_GPU");
        std::vector<int> ids((size_t)state.gpus);
        for (int i = 0; i < state.gpus; i++) ids[(size_t)i] = (reh && reh[0] == '1') ? state.device : state.device + i;
        if (pt_create_multi(&state.context, ids.data(), state.gpus) != 0) throw Exception(std::string("createDeviceContext: ") + pt_last_error(nullptr));
    } else if (pt_create(&state.context, state.device) != 0) throw Exception(std::string("createDeviceContext: ") + pt_last_error(nullptr));
    makeContextCurrent(state.context);      // OutputBuffer(type, w, h) allocates here, as CUDAOutputBuffer does on the current device
}

static void buildTheAccelarationStructure(PathTracerState& state, const TinyObjWrapper& objs)   // :260-398 (+ :544-627)
{
    std::vector<float> h_vertices = objs.getVerticesFloat();
    std::vector<uint32_t> h_mat_indices = objs.getMaterialIndices();
    std::vector<uint32_t> h_indxbuffer = objs.getIndexBuffer();
    std::vector<Material> materials = objs.getMaterials();
    PT_CHECK(state.context, pt_set_scene(state.context, h_vertices.data(), h_vertices.size() / 4, h_indxbuffer.data(), h_indxbuffer.size() / 3,
                                         h_mat_indices.data(), reinterpret_cast<const pt_material*>(materials.data()), materials.size()));
}

// The progressive state of the reference is the accumulation buffer and the frame index (pathTracerPrograms.cu:803-811).
// File: "ACGPTACC" | width | height | frames accumulated | float4[width * height]
static void saveAccumulation(PathTracerState& state, const std::string& path)
{
    const size_t n = (size_t)state.params.width * state.params.height * 4;
    std::vector<float> host(n);
    PT_CHECK(state.context, pt_copy_to_host(state.context, host.data(), state.params.accumulationBuffer, n * sizeof(float)));
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw Exception("cannot write " + path);
    const uint32_t hdr[3] = {state.params.width, state.params.height, state.params.currentFrameIdx};
    const bool ok = fwrite("ACGPTACC", 1, 8, f) == 8 && fwrite(hdr, 4, 3, f) == 3 && fwrite(host.data(), sizeof(float), n, f) == n;
    fclose(f);
    if (!ok) throw Exception("short write to " + path);
}
static void restoreAccumulation(PathTracerState& state, const std::string& path)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw Exception("cannot read " + path);
    char magic[8]; uint32_t hdr[3] = {0, 0, 0};
    const size_t n = (size_t)state.params.width * state.params.height * 4;
    std::vector<float> host(n);
    const bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "ACGPTACC", 8) == 0 && fread(hdr, 4, 3, f) == 3 &&
                    hdr[0] == state.params.width && hdr[1] == state.params.height && fread(host.data(), sizeof(float), n, f) == n;
    fclose(f);
    if (!ok) throw Exception(path + ": not an accumulation dump of a " + std::to_string(state.params.width) + "x" + std::to_string(state.params.height) + " image");
    PT_CHECK(state.context, pt_copy_to_device(state.context, state.params.accumulationBuffer, host.data(), n * sizeof(float)));
    state.params.currentFrameIdx = hdr[2];
}

// :400-627 — nothing to JIT, link or bind here (the gfx950 code object is built ahead of time, materials travel with pt_set_scene);
// the functions stay so that main() reads, and prints, like the reference's
static void createModule(PathTracerState&) {}
static void createProgramGroups(PathTracerState&) {}
static void createPipeline(PathTracerState&) {}
static void createShaderBindingTable(PathTracerState&, const TinyObjWrapper&) {}

static void CleanAllTheThings(PathTracerState& state)                    // :629-646
{
    if (state.params.accumulationBuffer) pt_device_free(state.context, state.params.accumulationBuffer);
    pt_destroy(state.context);
    state.context = nullptr;
}

int main(int argc, char** argv)
{
    std::string objfilepath, out = "frame.png", keys, save_accum, restore_accum;
    int32_t width = 512, height = 512, frames = 8, dump_every = 0;
    bool zero_copy = false;
    int orbit_dx = 0, orbit_dy = 0, zoom_steps = 0, sample_chunks = 0, build_mode = 1, fuse = 1, light_mode = 0, math_mode = PT_MATH_FAST;
    PathTracerState state;
    state.params.useDirectLighting = false;
    state.params.useImportanceSampling = false;
    state.params.maxDepth = 4;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; exit(2); } return argv[++i]; };
        if (a == "--obj") objfilepath = next();
        else if (a == "--width") width = atoi(next());
        else if (a == "--height") height = atoi(next());
        else if (a == "--spp-per-launch") samples_per_launch = atoi(next());
        else if (a == "--frames") frames = atoi(next());
        else if (a == "--max-depth") state.params.maxDepth = (uint32_t)std::min((int)maxiumumRecursionDepth, std::max(1, atoi(next())));
        else if (a == "--direct-lighting") state.params.useDirectLighting = true;
        else if (a == "--importance-sampling") state.params.useImportanceSampling = true;
        else if (a == "--device") state.device = atoi(next());
        else if (a == "--gpus") state.gpus = std::max(1, atoi(next()));
        else if (a == "--multi") state.multi = true;
        else if (a == "--save-accum") save_accum = next();
        else if (a == "--restore-accum") restore_accum = next();
        else if (a == "--keys") keys = next();
        else if (a == "--out") out = next();
        else if (a == "--dump-every") dump_every = atoi(next());
        else if (a == "--zero-copy") zero_copy = true;
        else if (a == "--orbit") { if (sscanf(next(), "%d,%d", &orbit_dx, &orbit_dy) != 2) { std::cerr << "--orbit dx,dy" << std::endl; return 2; } }
        else if (a == "--zoom") zoom_steps = atoi(next());
        else if (a == "--sample-chunks") sample_chunks = atoi(next());
        else if (a == "--build-mode") build_mode = atoi(next());
        else if (a == "--fuse-frames") fuse = std::min(64, std::max(1, atoi(next())));
        else if (a == "--math") { const std::string m = next(); math_mode = (m == "ieee" || m == "0") ? PT_MATH_IEEE : PT_MATH_FAST; }   // fast: the arithmetic of the reference's own build (nvcc --use_fast_math); ieee: the CPU oracle's
        else if (a == "--light-mode") light_mode = atoi(next());      // 0 = the reference's hard-coded rectangle (:154-158), 1 = the OBJ's emissive triangles + MIS
        else { std::cerr << "unknown option " << a << std::endl; return 2; }
    }
    if (objfilepath.empty()) { std::cerr << "usage: acgpt_main --obj scene.obj [options]" << std::endl; return 2; }
    std::vector<std::string> key_list;
    { std::stringstream ss(keys); std::string k; while (std::getline(ss, k, ',')) if (!k.empty()) key_list.push_back(k); }

    TinyObjWrapper obj(objfilepath);
    if (!obj.loaded()) return 1;
    state.params.width = width;
    state.params.height = height;
    try {
        initCamera();
        g_camera.setAspectRatio(static_cast<float>(state.params.width) / static_cast<float>(state.params.height));
        if (orbit_dx || orbit_dy || zoom_steps) {
            Trackball trackball;
            trackball.setCamera(&g_camera);
            trackball.setMoveSpeed(10.0f);
            trackball.setReferenceFrame(make_float3(1.0f, 0.0f, 0.0f), make_float3(0.0f, 0.0f, 1.0f), make_float3(0.0f, 1.0f, 0.0f));
            trackball.setGimbalLock(true);
            if (orbit_dx || orbit_dy) { trackball.startTracking(0, 0); trackball.updateTracking(orbit_dx, orbit_dy, width, height); }
            for (int z = 0; z < std::abs(zoom_steps); z++) trackball.wheelEvent(zoom_steps > 0 ? 1 : -1);
        }
        const float3 eye = g_camera.eye();
        state.params.cameraEye = {eye.x, eye.y, eye.z};
        float3 U, V, W;
        g_camera.UVWFrame(U, V, W);
        state.params.cameraU = {U.x, U.y, U.z}; state.params.cameraV = {V.x, V.y, V.z}; state.params.cameraW = {W.x, W.y, W.z};

        std::cout << "Using Direct Lighting: " << (state.params.useDirectLighting ? "yes" : "no") << std::endl;
        std::cout << "Using Importance Sampling: " << (state.params.useImportanceSampling ? "yes" : "no") << std::endl;
        createDeviceContext(state);
        PT_CHECK(state.context, pt_set_build_mode(state.context, build_mode));
        PT_CHECK(state.context, pt_set_sample_chunks(state.context, sample_chunks));
        PT_CHECK(state.context, pt_set_light_mode(state.context, light_mode));
        PT_CHECK(state.context, pt_set_math_mode(state.context, math_mode));
        buildTheAccelarationStructure(state, obj);
        std::cout << "Acceleration Structure Built" << std::endl;
        createModule(state);
        std::cout << "Module Created" << std::endl;
        createProgramGroups(state);
        std::cout << "Program Groups Created" << std::endl;
        createPipeline(state);
        std::cout << "Pipeline Created" << std::endl;
        createShaderBindingTable(state, obj);
        std::cout << "Shader Binding Table Created" << std::endl;
        initializeTheLaunch(state);
        std::cout << "Launch Initialized" << std::endl;
        if (!restore_accum.empty()) {
            restoreAccumulation(state, restore_accum);
            std::cout << "Accumulation restored: " << state.params.currentFrameIdx << " frames" << std::endl;
        }
        if (state.gpus > 1 || state.multi) std::cout << "Devices: " << pt_device_count(state.context) << std::endl;
        uint64_t rays = 0;
        {
            OutputBuffer<uchar4> output_buffer(zero_copy ? OutputBufferType::ZERO_COPY : OutputBufferType::DEVICE,
                                               state.params.width, state.params.height);
            size_t next_key = 0;
            for (int f = 0; f < frames;) {
                auto start = std::chrono::high_resolution_clock::now();
                if (next_key < key_list.size() && f > 0) { if (!keyCallback(state, key_list[next_key++])) break; }
                updateState(output_buffer, state);
                // a batch never crosses a key press or a dump point
                int batch = key_list.empty() ? std::min(fuse, frames - f) : 1;
                if (dump_every > 0) batch = std::min(batch, dump_every - f % dump_every);
                LaunchCurrentFrame(output_buffer, state, (uint32_t)batch);
                state.params.currentFrameIdx += (uint32_t)batch;
                f += batch - 1;
                auto end = std::chrono::high_resolution_clock::now();
                const double ms = std::chrono::duration<double, std::milli>(end - start).count();
                avg_ms += ms; total_ms += ms;
                sample_summ += samples_per_launch * batch;
                frame_counter += batch;
                pt_stats st; pt_get_stats(state.context, &st);
                rays += st.radiance_rays + st.shadow_rays;
                std::cout << "\rFrame Render Time: " << (long)ms << "ms" << std::flush;
                if (dump_every > 0 && (f + 1) % dump_every == 0 && f + 1 < frames) {
                    std::stringstream nm; nm << out << "." << (f + 1) << ".ppm";
                    saveImage(nm.str(), reinterpret_cast<const uint8_t*>(output_buffer.getHostPointer()), width, height);
                }
                f++;
            }
            std::cout << std::endl;
            if (!saveImage(out, reinterpret_cast<const uint8_t*>(output_buffer.getHostPointer()), width, height))
                std::cerr << "could not write " << out << std::endl;
            if (!save_accum.empty()) saveAccumulation(state, save_accum);
        }
        CleanAllTheThings(state);
        if (frame_counter > 0) avg_ms /= frame_counter;
        std::cout << "Total Samples " << sample_summ << std::endl;
        std::cout << "Average ms per frame: " << (long)avg_ms << std::endl;
        std::cout << "Total ms: " << (long)total_ms << std::endl;
        std::cout << "Rays: " << rays << "  Mray/s: " << (total_ms > 0 ? rays / total_ms / 1e3 : 0.0) << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "Caught exception: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
