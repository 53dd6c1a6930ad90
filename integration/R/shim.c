/* shim.c -- the only C a BayesLogit maintainer adds to Code/BLPackage/BayesLogit/src/ when the package's
 * shared object is libbayeslogit_hip.so.  It carries R's RNG state into the
 * library's counter RNG, so that set.seed() keeps governing the draws, as GetRNGstate()/PutRNGstate() do
 * in the reference entry points (Code/C/LogitWrapper.cpp:44-46,59-61,196-198,231-233).
 *
 * Not compiled in this repository (there is no R in the build image).
 */
#include <R.h>
#include <Rmath.h>

void bl_set_seed_from_unif(double *u);   /* include/bayeslogit_hip.h */

/* .C("bl_seed_from_R", PACKAGE = "BayesLogit") -- first line of every R wrapper that draws */
void bl_seed_from_R(void)
{
  double u[2];
  GetRNGstate();
  u[0] = unif_rand();
  u[1] = unif_rand();
  PutRNGstate();
  bl_set_seed_from_unif(u);
}
