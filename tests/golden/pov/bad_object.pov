camera { location <0,0,5> up <0,1,0> right <1.33,0,0> look_at <0,0,0> }
torus { <0,0,0>, 1 }
