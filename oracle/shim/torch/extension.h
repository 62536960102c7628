/* ORACLE shim: the kernel part of point_render.cu needs only the declaration in
 * point_render.cuh to parse; provide the two names it mentions. */
#ifndef ORACLE_SHIM_TORCH_EXTENSION_H
#define ORACLE_SHIM_TORCH_EXTENSION_H
#include <vector>
namespace torch { struct Tensor {}; }
#endif
