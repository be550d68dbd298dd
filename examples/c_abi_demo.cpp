// c_abi_demo.cpp -- the drop-in boundary used from plain C++ (no Python, no torch): load the compiled model, create a batched
// simulator on a HIP stream, put N humanoids in the standing pose with seeded perturbations, run control steps with zero controller
// action (stable-PD holds the pose, residual forces off) and print a checksum of the state.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_abi_demo.cpp -Lkinpoly_amd -lkinpoly_sim -Wl,-rpath,'$ORIGIN/../kinpoly_amd' -o examples/c_abi_demo
//   examples/c_abi_demo kinpoly_amd/assets/smpl_humanoid.kpm tests/golden/standing_neutral_qpos.f32 4096 10
//   examples/c_abi_demo --compile assets/mujoco_models/humanoid_smpl_neutral_mesh_all.xml config/uhc/uhc.yml out.kpm     (needs no GPU)
// The second form is mujoco_py.load_model_from_path's stand-in without Python: kp_model_compile reads the reference's XML, its STL meshes and
// the PD-gain table of uhc.yml and writes the blob the first form loads (a model argument ending in .xml is compiled on the spot).
//
// Every call below is an entry point of include/kinpoly_sim.h; errors come back as negative return codes + kp_last_error().
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "kinpoly_sim.h"

#define CK(x) do { if ((x) != 0) { std::fprintf(stderr, "%s failed: %s\n", #x, kp_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc >= 5 && std::string(argv[1]) == "--compile") {
        if (kp_model_compile(argv[2], argv[3][0] == '-' ? nullptr : argv[3], argv[4]) != 0) { std::fprintf(stderr, "kp_model_compile: %s\n", kp_last_error()); return 1; }
        std::printf("compiled %s -> %s\n", argv[2], argv[4]);
        return 0;
    }
    if (argc < 3) { std::fprintf(stderr, "usage: %s model.kpm standing_qpos.f32 [n_envs] [control_steps]\n", argv[0]); return 2; }
    const int n = argc > 3 ? std::atoi(argv[3]) : 4096, steps = argc > 4 ? std::atoi(argv[4]) : 10;
    std::vector<float> q0(KP_NQ);
    if (FILE* f = std::fopen(argv[2], "rb")) { size_t r = std::fread(q0.data(), sizeof(float), KP_NQ, f); std::fclose(f); if (r != KP_NQ) return 2; }
    else { std::perror(argv[2]); return 2; }

    const std::string mpath = argv[1];
    kp_model* model = mpath.size() > 4 && mpath.substr(mpath.size() - 4) == ".xml" ? kp_model_load_xml(argv[1], std::getenv("KP_UHC_YML")) : kp_model_load(argv[1]);
    if (!model) { std::fprintf(stderr, "kp_model_load: %s\n", kp_last_error()); return 1; }
    hipStream_t stream;
    HK(hipStreamCreate(&stream));
    kp_sim* sim = kp_sim_create(model, n, 0, stream);
    if (!sim) { std::fprintf(stderr, "kp_sim_create: %s\n", kp_last_error()); return 1; }

    // host-side initial state: standing pose + a small deterministic joint perturbation per env
    std::vector<float> qpos((size_t)n * KP_NQ), qvel((size_t)n * KP_NV, 0.f), act((size_t)n * KP_CC_ACTION_DIM, 0.f);
    uint32_t rng = 12345u;
    for (int e = 0; e < n; e++)
        for (int i = 0; i < KP_NQ; i++) {
            rng = rng * 1664525u + 1013904223u;
            const float u = (float)(rng >> 8) * (1.0f / 16777216.0f) - 0.5f;
            qpos[(size_t)e * KP_NQ + i] = q0[i] + (i >= 7 ? 0.05f * u : 0.f);
        }
    float *d_qpos, *d_qvel, *d_act, *d_out;
    HK(hipMalloc(&d_qpos, qpos.size() * 4)); HK(hipMalloc(&d_qvel, qvel.size() * 4)); HK(hipMalloc(&d_act, act.size() * 4)); HK(hipMalloc(&d_out, qpos.size() * 4));
    HK(hipMemcpy(d_qpos, qpos.data(), qpos.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_qvel, qvel.data(), qvel.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_act, act.data(), act.size() * 4, hipMemcpyHostToDevice));

    CK(kp_sim_set_state(sim, d_qpos, d_qvel, nullptr));      // set_state + sim.forward()
    CK(kp_sim_set_target(sim, d_qpos, nullptr));             // PD base pose = the initial pose
    for (int k = 0; k < steps; k++) CK(kp_sim_step_ctrl(sim, d_act, 15, nullptr));   // do_simulation(cc_action, 15)
    CK(kp_sim_get(sim, KP_QPOS, d_out));
    std::vector<int32_t> diag((size_t)n * 4);
    CK(kp_sim_diag(sim, diag.data()));                       // synchronises the stream
    std::vector<float> out(qpos.size());
    HK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));

    double sum = 0.0, zmin = 1e9; long contacts = 0; int bad = 0;
    for (int e = 0; e < n; e++) {
        for (int i = 0; i < KP_NQ; i++) sum += out[(size_t)e * KP_NQ + i];
        if (out[(size_t)e * KP_NQ + 2] < zmin) zmin = out[(size_t)e * KP_NQ + 2];
        contacts += diag[4 * e]; bad += diag[4 * e + 2] != 0;
    }
    std::printf("%s: %d envs x %d control steps, last launch %.3f ms, qpos checksum %.6f, lowest root height %.4f, contacts/env %.2f, non-finite %d\n",
                kp_version(), n, steps, kp_sim_last_step_seconds(sim) * 1e3, sum, zmin, (double)contacts / n, bad);
    kp_sim_destroy(sim); kp_model_free(model);
    (void)hipFree(d_qpos); (void)hipFree(d_qvel); (void)hipFree(d_act); (void)hipFree(d_out); (void)hipStreamDestroy(stream);
    return bad != 0;
}
