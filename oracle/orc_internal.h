/* TEST INFRASTRUCTURE (oracle) -- not part of the product.  See oracle/README.md. */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include <stdint.h>
#include "../include/lsim.h"
#include "../include/lsim_layout.h"

typedef struct orc_sim {
    lsim_config cfg;
    lsim_robot_model model;
    void* buf[LSIM_NUM_BUFFERS];
    int64_t step_counter;       /* common_step_counter, LR:948 */
    uint32_t reset_calls;       /* orc_reset_envs calls so far: salt of their random draws (include/lsim.h: lsim_reset_envs) */
    int init_done;              /* LR:97, LR:116 */
    int stats_row;              /* row of LSIM_BUF_STATS filled by the latest call */
    double command_ranges[4][2];/* python floats in the reference (LR:1256) */
    int active_terms[LSIM_NUM_REWARD_TERMS];  /* alphabetical, termination excluded (LR:1050-1055) */
    int num_active;
    char err[256];
    int8_t* vmove;              /* [rows*cols][2] horizontal displacement of each mesh vertex (orc_build_mesh_cache) */
    uint8_t* cell_walls;        /* [rows*cols] 1 where a displaced vertex lies in the 4x4 vertex block around the cell */
} orc_sim;

#define ORC_F(s, id) ((float*)(s)->buf[id])
#define ORC_U8(s, id) ((uint8_t*)(s)->buf[id])
#define ORC_I64(s, id) ((int64_t*)(s)->buf[id])
#define ORC_I32(s, id) ((int32_t*)(s)->buf[id])
#define ORC_I16(s, id) ((int16_t*)(s)->buf[id])

/* orc_physics.c: one 5 ms articulated-body sub-step of env `e` (the build's own solver; dynamics
 * parity with PhysX is unpinned -- SURVEY.md 8c).  tau: 12 joint efforts.  apply_force: consume pending_force. */
void orc_physics_substep(orc_sim* s, int e, const float tau[12], int apply_force);
/* recompute rigid_body_states of env e from root/dof state (forward kinematics + velocities) */
void orc_refresh_body_states(orc_sim* s, int e);
void orc_root_lin_vel_to_origin(orc_sim* s, int e);
/* terrain contact query: signed distance of a world point to the (slope-corrected) terrain mesh, and the contact normal */
void orc_terrain_contact(const orc_sim* s, const double cw[3], double radius, double* dist, double n[3]);
/* tabulate the vertex displacement rule once per terrain (called by orc_create after the grid is in place) */
void orc_build_mesh_cache(orc_sim* s);

#endif
