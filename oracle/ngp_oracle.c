/* ngp_oracle.c -- CPU oracle of the mapping path (added with the mapping kernels). */
