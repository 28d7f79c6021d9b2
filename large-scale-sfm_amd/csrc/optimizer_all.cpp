// optimizer_all -- the executable the pipeline's stage programs spawn after every triangulation and
// merge:  system("../bin/optimizer_all <dir>/ A B C D E F")   (SiftGPU/MergeClusters.cpp:1099-1110).
#include "../../include/mi_model.h"
int main(int argc, char **argv) { return mi_optimizer_all_main(argc, argv); }
