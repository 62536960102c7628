/* ORACLE — TEST INFRASTRUCTURE ONLY: serial stand-in for cooperative_groups::this_grid().
 * The driver sets the emulated thread id and the grid size before each kernel call. */
#ifndef ORACLE_SHIM_COOP_H
#define ORACLE_SHIM_COOP_H
namespace cooperative_groups {
struct grid_group {
    long long rank_, size_;
    long long thread_rank() const { return rank_; }
    long long size() const { return size_; }
};
extern thread_local grid_group oracle_current_grid;
static inline grid_group this_grid() { return oracle_current_grid; }
}
#endif
