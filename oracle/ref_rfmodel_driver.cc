// oracle/ref_rfmodel_driver.cc -- TEST INFRASTRUCTURE ONLY.
// Links the reference's dependency-free RF model I/O (ml/rf/ml_rf.h, ml/rf/ml_rf_model.cxx,
// compiled in place from /root/reference/code) to pin the model FILE FORMAT:
//   ref_rfmodel write <forest.txt> <model.bin>   build an rf_old::Model and write it with
//                                                 rf_old::writeModelToBinaryFile (ml_rf_model.cxx:378-455)
//   ref_rfmodel read  <model.bin>                read with rf_old::readModelFromBinaryFile (:459-563)
//                                                 and print the arrays the predictor uses
// forest.txt:  nrnodes ntree nclass mtry
//              orig_labels[nclass]  new_labels[nclass]
//              then per array (xbestsplit, treemap, nodestatus, nodeclass, bestvar, ndbigtree, classwt, cutoff):
//              n0 n1 followed by n0*n1 values in the writer's (pre-transpose) row-major order
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "ml/rf/ml_rf.h"

using namespace rf_old;

template <typename T> static T* readArr(FILE* f, int n[2], const char* fmt) {
  if (fscanf(f, "%d %d", &n[0], &n[1]) != 2) exit(2);
  int sz = n[0] * n[1];
  T* a = sz > 0 ? new T[sz] : NULL;
  for (int i = 0; i < sz; ++i) if (fscanf(f, fmt, &a[i]) != 1) exit(2);
  return a;
}
template <typename T> static void printArr(const char* name, T* a, int n[2], const char* fmt) {
  printf("%s %d %d", name, n[0], n[1]);
  for (int i = 0; i < n[0] * n[1]; ++i) { printf(" "); printf(fmt, a[i]); }
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "write")) {
    FILE* f = fopen(argv[2], "r");
    if (!f) return 2;
    Model m;
    if (fscanf(f, "%d %d %d %d", &m.nrnodes, &m.ntree, &m.nclass, &m.mtry) != 4) return 2;
    m.n_orig_labels[0] = m.nclass; m.n_orig_labels[1] = 1;
    m.n_new_labels[0] = m.nclass; m.n_new_labels[1] = 1;
    m.orig_labels = new int[m.nclass]; m.new_labels = new int[m.nclass];
    for (int i = 0; i < m.nclass; ++i) if (fscanf(f, "%d", &m.orig_labels[i]) != 1) return 2;
    for (int i = 0; i < m.nclass; ++i) if (fscanf(f, "%d", &m.new_labels[i]) != 1) return 2;
    m.xbestsplit = readArr<double>(f, m.n_xbestsplit, "%lf");
    m.treemap = readArr<int>(f, m.n_treemap, "%d");
    m.nodestatus = readArr<int>(f, m.n_nodestatus, "%d");
    m.nodeclass = readArr<int>(f, m.n_nodeclass, "%d");
    m.bestvar = readArr<int>(f, m.n_bestvar, "%d");
    m.ndbigtree = readArr<int>(f, m.n_ndbigtree, "%d");
    m.classwt = readArr<double>(f, m.n_classwt, "%lf");
    m.cutoff = readArr<double>(f, m.n_cutoff, "%lf");
    fclose(f);
    writeModelToBinaryFile(argv[3], m);
    return 0;
  }
  if (argc >= 3 && !strcmp(argv[1], "read")) {
    Model m;
    readModelFromBinaryFile(m, argv[2]);
    printf("sizeof_model %zu\n", sizeof(Model));
    printf("nrnodes %d ntree %d nclass %d mtry %d\n", m.nrnodes, m.ntree, m.nclass, m.mtry);
    printArr("orig_labels", m.orig_labels, m.n_orig_labels, "%d");
    printArr("new_labels", m.new_labels, m.n_new_labels, "%d");
    printArr("xbestsplit", m.xbestsplit, m.n_xbestsplit, "%.17g");
    printArr("treemap", m.treemap, m.n_treemap, "%d");
    printArr("nodestatus", m.nodestatus, m.n_nodestatus, "%d");
    printArr("nodeclass", m.nodeclass, m.n_nodeclass, "%d");
    printArr("bestvar", m.bestvar, m.n_bestvar, "%d");
    printArr("ndbigtree", m.ndbigtree, m.n_ndbigtree, "%d");
    printArr("classwt", m.classwt, m.n_classwt, "%.17g");
    printArr("cutoff", m.cutoff, m.n_cutoff, "%.17g");
    return 0;
  }
  fprintf(stderr, "usage: ref_rfmodel write <txt> <bin> | read <bin>\n");
  return 1;
}
