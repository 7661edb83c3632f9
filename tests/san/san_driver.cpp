// TEST INFRASTRUCTURE: runs the CPU lane emulator of the kernel sources (tests/emu) and the CPU oracle (oracle/) for a number of steps inside
// one process built with -fsanitize=address,undefined.  The configuration, robot model and terrain come in as raw blobs written by
// tests/test_sanitizers.py (the structs of include/lsim.h as the Python side fills them).  No checks of results here: the parity tests do
// that; this binary exists so that out-of-bounds LDS / buffer indexing, misaligned accesses and signed overflow in the shared phase
// code abort a test instead of going unnoticed.   usage: san_driver <cfg.bin> <model.bin> <grid.bin|-> <origins.bin|-> <steps> <actions.bin>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/lsim.h"

extern "C" {
int emu_create(const lsim_config*, const lsim_robot_model*, const int16_t*, const float*, void*, int, void**);
int emu_step_ex(void*, const float*, uint32_t, void*);
int emu_reset_all(void*, void*);
void emu_destroy(void*);
struct orc_sim;
int orc_create(const lsim_config*, const lsim_robot_model*, const int16_t*, const float*, orc_sim**);
int orc_step_ex(orc_sim*, const float*, uint32_t);
int orc_reset_all(orc_sim*);
void orc_destroy(orc_sim*);
}

static std::vector<char> slurp(const char* path) {
    std::vector<char> v;
    if (!strcmp(path, "-")) return v;
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage\n"); return 2; }
    auto cfgb = slurp(argv[1]), modb = slurp(argv[2]), grid = slurp(argv[3]), org = slurp(argv[4]), act = slurp(argv[6]);
    if (cfgb.size() != sizeof(lsim_config) || modb.size() != sizeof(lsim_robot_model)) { fprintf(stderr, "struct size mismatch\n"); return 3; }
    lsim_config cfg; lsim_robot_model model;
    memcpy(&cfg, cfgb.data(), sizeof(cfg)); memcpy(&model, modb.data(), sizeof(model));
    const int steps = atoi(argv[5]);
    const size_t per = (size_t)cfg.num_envs * 12 * sizeof(float);
    if (act.size() < per * (size_t)steps) { fprintf(stderr, "actions too short\n"); return 3; }
    const int16_t* g = grid.empty() ? nullptr : (const int16_t*)grid.data();
    const float* o = org.empty() ? nullptr : (const float*)org.data();
    void* e = nullptr; orc_sim* s = nullptr;
    if (emu_create(&cfg, &model, g, o, nullptr, 0, &e) != 0 || orc_create(&cfg, &model, g, o, &s) != 0) { fprintf(stderr, "create failed\n"); return 4; }
    if (emu_reset_all(e, nullptr) != 0 || orc_reset_all(s) != 0) return 5;
    for (int k = 0; k < steps; ++k) {
        const float* a = (const float*)(act.data() + per * (size_t)k);
        if (emu_step_ex(e, a, LSIM_STEP_DEFAULT, nullptr) != 0 || orc_step_ex(s, a, LSIM_STEP_DEFAULT) != 0) return 6;
    }
    emu_destroy(e); orc_destroy(s);
    printf("san_driver: %d steps x %d envs clean\n", steps, cfg.num_envs);
    return 0;
}
