// MergeClusters -- the stage program that fuses two sub-reconstructions with the best of the candidate transforms and has the
// result bundle adjusted (SiftGPU/MergeClusters.cpp:553-1115):  MergeClusters <output dir> <index>
#include "../../include/mi_merge.h"
int main(int argc, char **argv) { return mi_mergeclusters_main(argc, argv); }
