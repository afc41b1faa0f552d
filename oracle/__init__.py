"""oracle/ -- TEST INFRASTRUCTURE ONLY. CPU checker for the HIP path; never imported by the product package."""
