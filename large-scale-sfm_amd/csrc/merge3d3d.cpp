// Merge3D3D -- the stage program that fuses two sub-reconstructions and has the result bundle adjusted
// (SiftGPU/Merge3D3D.cpp:564-908):  Merge3D3D <output dir>
#include "../../include/mi_merge.h"
int main(int argc, char **argv) { return mi_merge3d3d_main(argc, argv); }
