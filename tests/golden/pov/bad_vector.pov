sphere { <0, 0>, 1 pigment { color rgb <1,1,1> } }
