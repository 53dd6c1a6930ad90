## The change to Code/R/LogitWrapper.R (INTEGRATION.md section 2): one line ahead of each .C call that draws.
## Everything else in the wrappers -- argument coercion, array(h, num) recycling, the transposes around
## gibbs / mult_gibbs, the N = OUT[[8]] read-back -- stays as it is.

## rpg (LogitWrapper.R:98-121)
    .C("bl_seed_from_R", PACKAGE="BayesLogit")
    OUT = .C("rpg_hybrid", x, h, z, as.integer(num), PACKAGE="BayesLogit")

## rpg.devroye :53-72, rpg.alt :76-95, rpg.sp :12-32, rpg.gamma :35-52: the same line before their .C call

## logit (LogitWrapper.R:197-244)
    .C("bl_seed_from_R", PACKAGE="BayesLogit")
    OUT = .C("gibbs", w, beta, as.double(y), as.double(t(X)), as.double(n), as.double(m0), as.double(P0),
             as.integer(N), as.integer(P), as.integer(samp), as.integer(burn), PACKAGE="BayesLogit")

## mlogit (LogitWrapper.R:323-368): the same line before .C("mult_gibbs", ...)

## optional: the unconstrained beta draw that Logit.hpp:430 comments out
    .C("bl_set_constrain_R", as.integer(0), PACKAGE="BayesLogit")
